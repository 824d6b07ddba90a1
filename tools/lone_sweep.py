"""GPU: one C2 ensemble at a time (and, with --fl, four in flight) under combinations of the cooperative-panel knobs.
`python tools/lone_sweep.py "T,budget,lane0_frac,lanes[,all[,old]]" ...`  e.g. "4,128,0.5,2" "8,256,0.6,2"."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llckbdm_amd import datasets                     # noqa: E402
from llckbdm_amd.engine import Engine                # noqa: E402


def lone(eng, sig, idx, m, reps=5):
    eng.solve(sig, idx, m, dwell=5e-4)
    best, st = 1e9, None
    for _ in range(reps):
        t0 = time.perf_counter()
        pend = eng.submit(sig, idx, m, dwell=5e-4)
        pend.result(check=False)
        dt = (time.perf_counter() - t0) * 1e3
        if dt < best:
            best, st = dt, pend.plan.stage_ms()
    return best, st


def in_flight(eng, works, steps, nfl):
    from collections import deque
    pend = deque()
    for w in works:
        eng.submit(*w, dwell=5e-4).result(check=False)
    t0 = time.perf_counter()
    for s in range(steps):
        if len(pend) == nfl:
            pend.popleft().result(check=False)
        pend.append(eng.submit(*works[s % nfl], dwell=5e-4))
    while pend:
        pend.popleft().result(check=False)
    return time.perf_counter() - t0


def main():
    fl = "--fl" in sys.argv
    combos = [a for a in sys.argv[1:] if not a.startswith("--")] or ["1,128,0.5,2", "4,128,0.5,2"]
    works = [datasets.config2(seed=1000 * k) for k in range(4)]
    sig, idx, m = works[0]
    for c in combos:
        p = c.split(",")
        os.environ["KBDM_PANEL_T"] = p[0]
        os.environ["KBDM_PANEL_BUDGET"] = p[1]
        os.environ["KBDM_LANE0_FRAC"] = p[2]
        os.environ["KBDM_LANES"] = p[3]
        os.environ["KBDM_PANEL_T_ALL"] = p[4] if len(p) > 4 else "0"
        eng = Engine(0, in_flight=4 if fl else 1)
        ms_, st = lone(eng, sig, idx, m)
        line = "%-22s lone %.2f ms (%.0f solves/s)  svd_fac %.2f gen %.2f hess %.2f hqr %.2f invit %.2f" % (
            c, ms_, len(m) / ms_ * 1e3, st["k_svd_fac"], st["k_gen(Q,P)"], st["k_hess"], st["k_hqr"], st["k_invit"])
        if fl:
            steps = 24
            dt = in_flight(eng, works, steps, 4)
            line += "  | 4 in flight %.0f solves/s" % (len(m) * steps / dt)
        print(line, flush=True)
        eng.close()


if __name__ == "__main__":
    main()
