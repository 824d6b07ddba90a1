#!/usr/bin/env python3
"""Round-2 golden vectors, produced by running the REFERENCE itself (build container only), as make_golden.py does:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r2.py

  n3m256q : a WELL-POSED Tikhonov case (reference kbdm.py:179-184): sigma = 1e-3 noise, m = 256, q = 1e-3
  c4m611  : a member of BASELINE.json config 4: N = 4096, the 16 table peaks + 16 seeded extra peaks
            (llckbdm_amd.datasets.config4), sigma = 1e-3 noise, m = 611
Outputs are DATA only (input signals, the arrays the reference returned, the analytic peak tables)."""
import os
import sys

import numpy as np

np.complex = complex  # alias removed in numpy>=1.24; reference kbdm.py:111-113 needs it

REF = os.environ.get("LLCKBDM_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import pandas as pd  # noqa: E402
from llckbdm import sig_gen  # noqa: E402
from llckbdm.kbdm import kbdm  # noqa: E402
from llckbdm.sampling import filter_samples  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
DWELL = 5e-4


def canonical(ll):
    with np.errstate(all="ignore"):
        return ll[np.lexsort((1.0 / ll[:, 1], ll[:, 2]))]


def noisy(sig, sigma, seed):
    rng = np.random.default_rng(seed)
    n = rng.standard_normal(sig.shape[0]) + 1j * rng.standard_normal(sig.shape[0])
    return sig + sigma * n


def main():
    out = {}
    df = pd.read_csv(f"{REF}/data/params_brain_sim_1_5T.csv",
                     names=["amplitude", "t2", "frequency", "phase"]).sort_values(["frequency"])
    params16 = df.values
    t = np.linspace(0, DWELL * 2048, 2048, endpoint=False)
    out["sig2048_n3"] = noisy(sig_gen.multi_fid(t, params16), 1e-3, 0)
    # config 4: 16 extra peaks from default_rng(1), t = arange(N) * dwell, noise seed 4
    rng = np.random.default_rng(1)
    extra = np.column_stack([rng.uniform(0.01, 1, 16), rng.uniform(0.005, 0.2, 16), rng.uniform(50, 950, 16), np.zeros(16)])
    params32 = np.vstack([params16, extra])
    out["params32"] = params32
    out["sig4096_c4"] = noisy(sig_gen.multi_fid(np.arange(4096) * DWELL, params32), 1e-3, 4)
    cases = [("n3m256q", "sig2048_n3", 256, None, 1, 1e-3), ("c4m611", "sig4096_c4", 611, None, 1, 0.0)]
    names = []
    for name, key, m, l, p, q in cases:
        ll, info = kbdm(out[key], DWELL, m=m, p=p, l=l, q=q)
        names.append(name)
        out[f"{name}__meta"] = np.array([m, info.l, p], dtype=np.int64)
        out[f"{name}__q"] = np.array([q], dtype=np.float64)
        out[f"{name}__sig"] = np.array(key)
        out[f"{name}__kept"] = canonical(filter_samples(ll))
        out[f"{name}__sv"] = np.asarray(info.singular_values)
        print(f"{name:8s} m={m:4d} l={info.l:4d} p={p} q={q:g} kept={len(filter_samples(ll))}")
    out["case_names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "kbdm_golden_r2.npz"), **out)
    print("wrote", os.path.join(HERE, "kbdm_golden_r2.npz"))


if __name__ == "__main__":
    main()
