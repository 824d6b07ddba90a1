import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.linalg as sl
from llckbdm_amd import datasets
from llckbdm_amd.engine import Engine
eng = Engine(0, in_flight=1)
sigs, sidx, ms = datasets.config4()
sig = sigs[0]
def hank(m): return sl.hankel(sig[:m], sig[m-1:2*m-1])
for batch in ([200, 611], [200, 300], [200, 513], [200, 150]):
    a, _ = eng.svd([hank(m) for m in batch]); b, _ = eng.svd([hank(200)])
    print("svd stage hankel m=200 with", batch[1], ": s", np.abs(a[0][1]-b[0][1]).max(), "L", np.abs(a[0][0]-b[0][0]).max(), "R", np.abs(a[0][2]-b[0][2]).max(), flush=True)
for batch in ([200, 611], [200, 300], [200, 513], [200, 150], [200, 1200]):
    r = eng.solve(sigs, [0, 0], batch, None, dwell=datasets.DWELL)
    s = eng.solve(sigs, [0], [200], None, dwell=datasets.DWELL)
    print("pipeline m=200 with", batch[1], ": sv", np.abs(r.singular_values(0)-s.singular_values(0)).max(), "mu", np.abs(r.eigenvalues(0)-s.eigenvalues(0)).max(),
          "lines", np.abs(r.line_list(0)-s.line_list(0)).max(), "status", r.status, s.status, flush=True)
