#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE itself (build container only).

Usage (from the repo root, in the build container where /root/reference exists):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference (danilomendesdias/llckbdm v0.2.4, read-only at /root/reference) is imported,
never copied.  ``np.complex`` was removed in numpy >= 1.24 and the reference still uses it
(kbdm.py:111-113), so this harness restores the alias in its own process before calling.
Outputs are DATA only: input signals and the arrays the reference returned for them.
Row order of ``kbdm`` output follows LAPACK zgeev's eigenvalue order, so line lists are
stored both raw and canonicalised (sorted by frequency, then 1/T2).
"""
import os
import sys

import numpy as np

np.complex = complex  # alias removed in numpy>=1.24; reference kbdm.py:111-113 needs it

REF = os.environ.get("LLCKBDM_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import pandas as pd  # noqa: E402
from llckbdm import sig_gen  # noqa: E402
from llckbdm.kbdm import kbdm, _compute_U_matrices  # noqa: E402
from llckbdm.sampling import sample_kbdm, filter_samples  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
DWELL = 5e-4


def canonical(ll):
    with np.errstate(all="ignore"):
        return ll[np.lexsort((1.0 / ll[:, 1], ll[:, 2]))]


def ref_signal(N):
    # exactly the reference fixture: _tests/fixtures.py:9-47
    df = pd.read_csv(f"{REF}/data/params_brain_sim_1_5T.csv",
                     names=["amplitude", "t2", "frequency", "phase"]).sort_values(["frequency"])
    t = np.linspace(0, DWELL * N, N, endpoint=False)
    return sig_gen.multi_fid(t, df.values), df.values


def noisy(sig, sigma, seed):
    rng = np.random.default_rng(seed)
    n = rng.standard_normal(sig.shape[0]) + 1j * rng.standard_normal(sig.shape[0])
    return sig + sigma * n


def main():
    out = {}
    sig2048, params = ref_signal(2048)
    sig1024, _ = ref_signal(1024)
    out["params_sorted"] = params
    out["sig2048"] = sig2048
    out["sig1024"] = sig1024
    sig_n3 = noisy(sig2048, 1e-3, 0)      # C2-style input
    sig_n6 = noisy(sig2048, 1e-6, 7)      # pseudo-noise style input
    out["sig2048_n3"] = sig_n3
    out["sig2048_n6"] = sig_n6

    cases = [
        # name,     signal key,   m,    l,   p, q
        ("c1",      "sig1024",    300, None, 1, 0),
        ("m300",    "sig2048",    300, None, 1, 0),
        ("m150",    "sig2048",    150, None, 1, 0),
        ("m100",    "sig2048",    100, None, 1, 0),
        ("m101",    "sig2048",    101, None, 1, 0),
        ("m102",    "sig2048",    102, None, 1, 0),
        ("m30",     "sig2048",     30,   30, 1, 0),
        ("m10q",    "sig2048",     10, None, 1, 1e-3),
        ("m180l30", "sig2048",    180,   30, 1, 0),
        ("m64p2",   "sig2048",     64, None, 2, 0),
        ("n3m128",  "sig2048_n3", 128, None, 1, 0),
        ("n3m256",  "sig2048_n3", 256, None, 1, 0),
        ("n6m256",  "sig2048_n6", 256, None, 1, 0),
        ("n3m512",  "sig2048_n3", 512, None, 1, 0),
    ]
    names = []
    for name, key, m, l, p, q in cases:
        ll, info = kbdm(out[key], DWELL, m=m, p=p, l=l, q=q)
        names.append(name)
        out[f"{name}__meta"] = np.array([m, info.l, p], dtype=np.int64)
        out[f"{name}__q"] = np.array([q], dtype=np.float64)
        out[f"{name}__sig"] = np.array(key)
        out[f"{name}__raw"] = ll
        out[f"{name}__canon"] = canonical(ll)
        out[f"{name}__kept"] = canonical(filter_samples(ll))
        out[f"{name}__sv"] = np.asarray(info.singular_values)
        print(f"{name:8s} m={m:4d} l={info.l:4d} p={p} q={q:g} kept={len(filter_samples(ll))}")
    out["case_names"] = np.array(names)

    # Hankel rows (reference test_kbdm.py:45-59 checks first/last rows for p=2, m=300)
    U0, Up_1, Up = _compute_U_matrices(data=sig2048, m=300, p=2)
    out["hankel_p2_m300_U0_rows"] = np.stack([U0[0], U0[-1]])
    out["hankel_p2_m300_Up1_rows"] = np.stack([Up_1[0], Up_1[-1]])
    out["hankel_p2_m300_Up_rows"] = np.stack([Up[0], Up[-1]])
    U0, Up_1, Up = _compute_U_matrices(data=sig2048, m=17, p=3)
    out["hankel_p3_m17_U0"], out["hankel_p3_m17_Up1"], out["hankel_p3_m17_Up"] = U0, Up_1, Up

    # sampler (reference test_sampling.py:19-49): m=100..102, unfiltered and filtered
    lls, infos = sample_kbdm(sig2048, DWELL, range(100, 103), p=1, l=None, q=0,
                             filter_invalid_features=True)
    out["sample_100_103_counts"] = np.array([len(x) for x in lls])
    out["sample_100_103_ms"] = np.array([i.m for i in infos])
    for i, x in enumerate(lls):
        out[f"sample_100_103_ll{i}"] = canonical(x)

    np.savez_compressed(os.path.join(HERE, "kbdm_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "kbdm_golden.npz"))


if __name__ == "__main__":
    main()
