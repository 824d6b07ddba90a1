"""GPU vs oracle for tiny members (m = 1 .. 9, p = 1 .. 3, l <= m) and a pure-noise signal."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.helpers import canonical, keep_mask
from llckbdm_amd.engine import Engine
from llckbdm_amd.kbdm import kbdm
from oracle import kbdm_oracle as O
eng = Engine(0)
rng = np.random.default_rng(1)
n = np.arange(64)
sig = 2.0 * np.exp((-0.01 + 0.3j) * n) + 0.7 * np.exp((-0.03 - 1.1j) * n) + 1e-3 * (rng.standard_normal(64) + 1j * rng.standard_normal(64))
worst = 0.0
for m in (1, 2, 3, 4, 5, 7, 9):
    for p in (1, 2, 3):
        for l in sorted({m, max(1, m - 1), max(1, m // 2)}):
            try:
                ll, info = kbdm(sig, 5e-4, m=m, p=p, l=l, engine=eng)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    want, _ = O.kbdm(sig, 5e-4, m=m, p=p, l=l)
                a, b = canonical(ll), canonical(want)
                strong = np.abs(b[:, 0]) > 1e-3
                d = np.abs(a[strong][:, :3] - b[strong][:, :3]) / np.maximum(np.abs(b[strong][:, :3]), 1e-300)
                e = float(d.max()) if d.size else 0.0
                worst = max(worst, e)
                if e > 1e-8 or a.shape != b.shape:
                    print("m", m, "p", p, "l", l, "dev", e, a.shape, b.shape)
            except Exception as ex:
                print("m", m, "p", p, "l", l, type(ex).__name__, str(ex)[:80])
print("worst deviation of the lines with A > 1e-3:", worst)
noise = rng.standard_normal(512) + 1j * rng.standard_normal(512)
ll, info = kbdm(noise, 5e-4, m=128, p=1, engine=eng)
want, _ = O.kbdm(noise, 5e-4, m=128, p=1)
a, b = canonical(ll[keep_mask(ll)]), canonical(want[keep_mask(want)])
print("pure noise m=128: kept", len(a), len(b), "max rel dev", float((np.abs(a[:, :3] - b[:, :3]) / np.abs(b[:, :3])).max()) if a.shape == b.shape else "shape")
