"""Soak run (GPU box): many ensembles with different noise seeds and size ranges; every member's status word must be zero and
the eigenvalues of a few members per ensemble are compared with numpy's (checker use only).  Looks for rare protocol /
geometry failures of the kind the C4 member m = 1123 showed."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llckbdm_amd import datasets, sig_gen
from llckbdm_amd.engine import Engine
from oracle import kbdm_oracle as O

eng = Engine(0)
bad_total = 0
t0 = time.time()


def check(name, sigs, sig_idx, ms, oracle_idx=()):
    global bad_total
    res = eng.solve(sigs, sig_idx, ms, None, p=1, q=0.0, dwell=datasets.DWELL)
    bad = np.nonzero(res.status)[0]
    worst = 0.0
    for i in oracle_idx:
        m = int(ms[i])
        want, info, mu_ref = O.kbdm(sigs[sig_idx[i]], datasets.DWELL, m=m, normalizer="gemm", return_mu=True)
        mu = res.eigenvalues(i)
        d = np.abs(mu[:, None] - np.asarray(mu_ref)[None, :]).min(axis=1)
        worst = max(worst, float(d.max()))
    print(f"{name}: members {len(ms)} non-zero status {[(int(ms[i]), int(res.status[i])) for i in bad]} eig dist {worst:.2e} t={time.time() - t0:.0f}s", flush=True)
    bad_total += len(bad) + (1 if worst > 1e-9 else 0)


nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for seed in range(nseeds):
    sigs, si, ms = datasets.config2(seed=seed)
    check(f"C2 seed {seed}", sigs, si, ms, oracle_idx=(150, 75) if seed < 10 else ())
for seed in range(4):
    sigs, si, ms = datasets.north_star(seed=100 + seed)
    check(f"NS seed {seed}", sigs, si, ms, oracle_idx=(200,))
# odd sizes and the large members in different batch compositions
for lo, hi, st in ((201, 1199, 7), (600, 1200, 13), (1000, 1200, 3), (64, 700, 5)):
    sigs, si, ms = datasets.config4(mmin=lo, mmax=hi, step=st)
    check(f"C4 m={lo}..{hi}:{st}", sigs, si, ms, oracle_idx=(len(ms) - 1,))
print("TOTAL failures", bad_total)
sys.exit(1 if bad_total else 0)
