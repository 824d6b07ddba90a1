"""GPU: cooperative panels (kb_team.hpp).  The C2 ensemble with teams of 1, 2, 4, 8 workgroups per member of lane 0 must give
the same BITS; prints the stage timers of lane 0 for one ensemble at a time.  `python tools/check_team.py [T ...]`"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llckbdm_amd import datasets                     # noqa: E402
from llckbdm_amd.engine import Engine                # noqa: E402


def run(T, budget, sig, m, reps=3):
    os.environ["KBDM_PANEL_T"] = str(T)
    os.environ["KBDM_PANEL_BUDGET"] = str(budget)
    eng = Engine(0, in_flight=1)
    eng.wide_solve = False
    idx = np.zeros(len(m), np.int32)
    res = eng.solve(sig, idx, m, dwell=5e-4)
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        pend = eng.submit(sig, idx, m, dwell=5e-4)
        r2 = pend.result(check=False)
        dt = (time.perf_counter() - t0) * 1e3
        st = pend.plan.stage_ms()
        if best is None or dt < best[0]:
            best = (dt, st)
    eng.close()
    return res, best


def main():
    Ts = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
    sig, _, m = datasets.config2()
    ref, best = run(1, 256, sig, m)
    base = None
    for T in Ts:
        res, best = run(T, 256, sig, m)
        if base is None:
            base = res
        same = all(np.array_equal(getattr(res, k), getattr(base, k)) for k in ("lines", "sv", "mu", "keep", "status"))
        dev = np.abs(res.sv - ref.sv).max() / ref.sv.max()
        print("T=%d: %.2f ms  svd_fac %.2f  hess %.2f  hqr %.2f  bitwise==T%d: %s  sv vs a team of one: %.1e  status!=0: %d" %
              (T, best[0], best[1]["k_svd_fac"], best[1]["k_hess"], best[1]["k_hqr"], Ts[0], same, dev,
               int(np.count_nonzero(res.status))), flush=True)


if __name__ == "__main__":
    main()
