"""GPU vs oracle on degenerate signals: all zero, a constant, one exponential (rank 1), two exponentials."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from helpers import canonical, keep_mask
from llckbdm_amd.engine import Engine
from llckbdm_amd.kbdm import kbdm
from oracle import kbdm_oracle as O
eng = Engine(0)
N = 256; n = np.arange(N)
cases = {"zeros": np.zeros(N, complex), "constant": np.ones(N, complex),
         "one exponential": 2.0 * np.exp((-0.01 + 0.3j) * n),
         "two exponentials": 2.0 * np.exp((-0.01 + 0.3j) * n) + 0.5 * np.exp((-0.02 - 0.7j) * n + 0.4j)}
for name, sig in cases.items():
    for m in (12, 64):
        out = []
        for lab, fn in (("oracle", lambda: O.kbdm(sig, 5e-4, m=m, p=1, l=None, q=0)), ("gpu", lambda: kbdm(sig, 5e-4, m=m, p=1, l=None, q=0, engine=eng))):
            try:
                with warnings.catch_warnings(record=True) as w:
                    warnings.simplefilter("always")
                    ll, info = fn()
                k = ll[keep_mask(ll)] if np.isfinite(ll).any() else ll[:0]
                out.append(f"{lab}: finite {bool(np.isfinite(ll).all())} kept {len(k)} strongest {np.sort(np.nan_to_num(ll[:, 0]))[-2:]} warnings {sorted(set(type(x.message).__name__ for x in w))}")
            except Exception as e:
                out.append(f"{lab}: raises {type(e).__name__}: {str(e)[:70]}")
        print(f"{name:18s} m={m:3d} | " + " | ".join(out))
