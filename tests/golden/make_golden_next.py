#!/usr/bin/env python3
"""Golden vectors for the rows after the hot path: frequency-domain RMSE scoring
(`metrics.calculate_freq_domain_rmse`, metrics.py:7-17) and `min_rmse_kbdm` (min_rmse_kbdm.py:21-55),
produced by running the REFERENCE itself (build container only; same harness rules as make_golden.py:
imported, never copied; `np.complex` alias restored for kbdm.py:111-113).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_next.py

`llckbdm/llckbdm.py` cannot be imported here (it imports the absent `hdbscan` package at line 3), so the
clustering sweep has no reference-generated vectors: its pure-numpy helpers are pinned by known-answer
tests that follow the reference's own tests (_tests/test_llckbdm.py:11-37), cluster labels are unpinned.
"""
import os
import sys

import numpy as np

np.complex = complex

REF = os.environ.get("LLCKBDM_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import pandas as pd  # noqa: E402
from llckbdm import sig_gen  # noqa: E402
from llckbdm.metrics import calculate_freq_domain_rmse  # noqa: E402
from llckbdm.min_rmse_kbdm import min_rmse_kbdm  # noqa: E402
from llckbdm.sampling import sample_kbdm  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
DWELL = 5e-4


def ref_signal(N):
    df = pd.read_csv(f"{REF}/data/params_brain_sim_1_5T.csv",
                     names=["amplitude", "t2", "frequency", "phase"]).sort_values(["frequency"])
    t = np.linspace(0, DWELL * N, N, endpoint=False)
    return sig_gen.multi_fid(t, df.values), df.values


def main():
    out = {}
    sig, params = ref_signal(2048)
    rng = np.random.default_rng(7)
    noisy = sig + 1e-2 * (rng.standard_normal(2048) + 1j * rng.standard_normal(2048))
    out["sig2048"] = sig
    out["noisy2048"] = noisy
    out["params"] = params
    # --- RMSE scoring: candidates of several sizes against clean and noisy data, N = 2048 and N = 1000
    # (gen_t_freq_arrays builds t with np.arange(0, N*dwell, dwell), sig_gen.py:20: for some N, e.g. 1001, that
    # yields N+1 points and the reference raises inside sklearn; such N are outside the contract)
    cands = [params, params[:5], params[3:4]]
    pert = params.copy()
    pert[:, 0] *= 1.0 + 0.05 * rng.standard_normal(len(pert))
    pert[:, 2] += 0.3 * rng.standard_normal(len(pert))
    pert[:, 3] = 0.2 * rng.standard_normal(len(pert))
    cands.append(pert)
    big = np.column_stack([rng.uniform(0.01, 1.0, 300), rng.uniform(0.005, 0.2, 300),
                           rng.uniform(-900, 900, 300), rng.uniform(-3, 3, 300)])
    cands.append(big)
    out["rmse_ncand"] = np.array([len(cands)])
    for i, c in enumerate(cands):
        out[f"rmse_cand{i}"] = c
        out[f"rmse_clean{i}"] = np.array([calculate_freq_domain_rmse(data=sig, params_est=c, dwell=DWELL)])
        out[f"rmse_noisy{i}"] = np.array([calculate_freq_domain_rmse(data=noisy, params_est=c, dwell=DWELL)])
    sig_odd = sig[:1000]
    out["sig1000"] = sig_odd
    out["rmse_odd"] = np.array([calculate_freq_domain_rmse(data=sig_odd, params_est=pert, dwell=DWELL)])
    # --- min_rmse_kbdm: the reference's own test case (_tests/test_min_rmse_kbdm.py:6-23)
    m_range = [30, 31, 180, 32, 33, 34]
    res = min_rmse_kbdm(data=sig, dwell=DWELL, m_range=m_range, l=30)
    out["minrmse_m_range"] = np.array(m_range)
    out["minrmse_rmses"] = np.array(res.rmses_list)
    out["minrmse_index"] = np.array([res.min_index])
    out["minrmse_min"] = np.array([res.min_rmse])
    out["minrmse_counts"] = np.array([len(s) for s in res.samples])
    # and on the noisy signal, where no candidate is exact
    samples, _ = sample_kbdm(data=noisy, dwell=DWELL, m_range=range(100, 110), p=1, l=None, q=0)
    res2 = min_rmse_kbdm(data=noisy, dwell=DWELL, samples=samples)
    out["minrmse2_rmses"] = np.array(res2.rmses_list)
    out["minrmse2_index"] = np.array([res2.min_index])
    np.savez_compressed(os.path.join(HERE, "next_golden.npz"), **out)
    print("wrote next_golden.npz:", {k: v.shape for k, v in out.items() if k.startswith("rmse_c") or k.startswith("minrmse")})
    print("rmses", res.rmses_list, res.min_index, res2.rmses_list, res2.min_index)


if __name__ == "__main__":
    main()
