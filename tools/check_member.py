"""One member of a config against the oracle: python tools/check_member.py C4 1123  (checker use of the oracle only)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llckbdm_amd import datasets
from llckbdm_amd.engine import Engine
from oracle import kbdm_oracle as O
from tests.helpers import canonical, keep_mask
cfg, m = sys.argv[1], int(sys.argv[2])
sigs = {"C4": datasets.config4, "C2": datasets.config2}[cfg]()[0]
eng = Engine(0)
res = eng.solve(sigs, [0], [m], None, p=1, q=0.0, dwell=datasets.DWELL)
want, info, mu_ref = O.kbdm(sigs[0], datasets.DWELL, m=m, normalizer="gemm", return_mu=True) if "return_mu" in O.kbdm.__code__.co_varnames else (*O.kbdm(sigs[0], datasets.DWELL, m=m, normalizer="gemm"), None)
got = res.line_list(0)
k, w = canonical(got[keep_mask(got)]), canonical(O.filter_samples(want))
print("m", m, "status", int(res.status[0]), "kept", len(k), len(w))
if len(k) == len(w):
    rel = np.abs(k[:, :3] - w[:, :3]) / np.abs(w[:, :3])
    strong = w[:, 0] > 1e-4
    print("  worst rel (A>1e-4): %.3e  all: %.3e" % (rel[strong].max(), rel.max()))
if mu_ref is not None:
    mu = res.eigenvalues(0)
    d = np.abs(mu[:, None] - np.asarray(mu_ref)[None, :]).min(axis=1)
    print("  eigenvalue distance to the oracle's: max %.3e, count > 1e-10: %d" % (d.max(), int((d > 1e-10).sum())))
