"""GPU: is a member's result independent of the batch it is solved in (divide-and-conquer SVD)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from llckbdm_amd import datasets
from llckbdm_amd.engine import Engine
eng = Engine(0, in_flight=1)
sigs, sidx, ms = datasets.config4()
sel = np.array([0, 1, 50, 411, 700, 1000])
for trial in range(2):
    res = eng.solve(sigs, sidx[sel], ms[sel], None, dwell=datasets.DWELL)
    full = eng.solve(sigs, sidx[::10], ms[::10], None, dwell=datasets.DWELL)
    for k, i in enumerate(sel):
        solo = eng.solve(sigs, [0], [int(ms[i])], None, dwell=datasets.DWELL)
        dsv = np.abs(solo.singular_values(0) - res.singular_values(k)).max()
        dl = np.abs(solo.line_list(0) - res.line_list(k)).max()
        extra = ""
        if i % 10 == 0:
            j = i // 10
            extra = f" | vs stride-10 batch: sv {np.abs(solo.singular_values(0) - full.singular_values(j)).max():.2e} lines {np.abs(solo.line_list(0) - full.line_list(j)).max():.2e}"
        print(f"trial {trial} m={ms[i]} solo vs small batch: sv {dsv:.2e} lines {dl:.2e}{extra}", flush=True)
# svd stage alone
rng = np.random.default_rng(0)
mats = [rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)) for n in (200, 64, 700)]
a, st = eng.svd(mats)
b, st2 = eng.svd(mats[:1])
print("svd stage m=200 in batch vs solo: s", np.abs(a[0][1] - b[0][1]).max(), "L", np.abs(a[0][0] - b[0][0]).max(), "R", np.abs(a[0][2] - b[0][2]).max())
c, _ = eng.svd(mats)
print("svd stage repeat: ", max(np.abs(x[0] - y[0]).max() for x, y in zip(a, c)))
