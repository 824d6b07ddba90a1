"""C4 at full size: members whose status word is non-zero, compared with the oracle (checker use only)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llckbdm_amd import datasets
from llckbdm_amd.engine import Engine
from oracle import kbdm_oracle as O
from tests.helpers import canonical, keep_mask
eng = Engine(0)
sigs, sig_idx, ms = datasets.config4()
res = eng.solve(sigs, sig_idx, ms, None, p=1, q=0.0, dwell=datasets.DWELL)
bad = np.nonzero(res.status)[0]
print("non-zero status:", [(int(ms[i]), int(res.status[i])) for i in bad])
for i in bad[:3]:
    m = int(ms[i])
    want, info = O.kbdm(sigs[0], datasets.DWELL, m=m, normalizer="gemm")
    got = res.line_list(i)
    k, w = canonical(got[keep_mask(got)]), canonical(O.filter_samples(want))
    print("m", m, "kept", len(k), len(w))
    if len(k) == len(w):
        rel = np.abs(k[:, :3] - w[:, :3]) / np.abs(w[:, :3])
        strong = w[:, 0] > 1e-4
        print("  worst rel (A>1e-4):", rel[strong].max(), " all:", rel.max())
