"""Throughput of the two kernels next to the hot path (run on the GPU box): candidates/s of the RMSE scorer and
samples^2/s of the silhouette kernel, with the CPU restatement (oracle / scikit-learn) timed beside them on a
bounded sample.  Prints one JSON line per kernel."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llckbdm_amd import datasets  # noqa: E402
from llckbdm_amd.engine import Engine  # noqa: E402
from oracle import llc_oracle as L  # noqa: E402  (CPU baseline only)

eng = Engine(0)
rng = np.random.default_rng(0)
sigs, _, _ = datasets.config2(seed=0)
sig = sigs[0]
dwell = datasets.DWELL
# ---- RMSE: 151 candidates of 100..400 lines (what min_rmse_kbdm scores after a C2 ensemble)
cands = [np.column_stack([rng.uniform(0.01, 1, k), rng.uniform(0.005, 0.2, k), rng.uniform(-900, 900, k), rng.uniform(-3, 3, k)])
         for k in range(100, 402, 2)]
eng.rmse_batch(sig, dwell, cands)
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    got = eng.rmse_batch(sig, dwell, cands)
gpu_s = (time.perf_counter() - t0) / reps
t0 = time.perf_counter()
ref = [L.calculate_freq_domain_rmse(sig, c, dwell) for c in cands[:20]]
cpu_s = (time.perf_counter() - t0) / 20 * len(cands)
terms = sum(len(c) for c in cands) * len(sig)
print(json.dumps({"kernel": "k_rmse", "candidates": len(cands), "N": len(sig), "gpu_ms_incl_transfers": 1e3 * gpu_s,
                  "gpu_terms_per_s": terms / gpu_s, "cpu_oracle_ms_extrapolated": 1e3 * cpu_s,
                  "max_rel_err_vs_oracle": float(np.max(np.abs(got[:20] - ref) / np.array(ref)))}))
# ---- silhouettes: 19k pooled lines in 4-d, 30 clusters + noise (C2 scale)
n = 19000
centres = rng.standard_normal((30, 4))
lab = rng.integers(-1, 30, n)
X = np.where(lab[:, None] >= 0, centres[np.clip(lab, 0, 29)] + 1e-3 * rng.standard_normal((n, 4)), rng.standard_normal((n, 4)))
eng.silhouette_samples(X, lab)
t0 = time.perf_counter()
for _ in range(reps):
    s_gpu = eng.silhouette_samples(X, lab)
gpu_s = (time.perf_counter() - t0) / reps
from sklearn.metrics import silhouette_samples  # noqa: E402
t0 = time.perf_counter()
s_cpu = silhouette_samples(X, lab)
cpu_s = time.perf_counter() - t0
print(json.dumps({"kernel": "k_silhouette", "n": n, "dim": 4, "classes": 31, "gpu_ms_incl_transfers": 1e3 * gpu_s,
                  "gpu_pairs_per_s": n * n / gpu_s, "cpu_sklearn_ms": 1e3 * cpu_s,
                  "max_abs_diff_vs_sklearn": float(np.abs(s_gpu - s_cpu).max())}))

# ---- HDBSCAN sweep: the 150 fits of one llc_kbdm call on the pooled lines of a C2 ensemble
ks = list(range(1, 151))
eng.hdbscan_sweep(X[:2000], ks[:4])
t0 = time.perf_counter()
labels, ncl = eng.hdbscan_sweep(X, ks)
gpu_s = time.perf_counter() - t0
from sklearn.cluster import HDBSCAN  # noqa: E402
from sklearn.metrics import adjusted_rand_score  # noqa: E402
t0 = time.perf_counter()
refs = {k: HDBSCAN(min_samples=k, min_cluster_size=5, copy=True).fit(X).labels_ for k in (1, 50, 150)}
cpu_s = (time.perf_counter() - t0) / 3 * len(ks)
print(json.dumps({"kernel": "k_knn_dist + k_prim_mst + host trees", "n": n, "fits": len(ks), "gpu_s_incl_transfers": gpu_s,
                  "cpu_sklearn_s_extrapolated": cpu_s,
                  "clusters_gpu_vs_sklearn": {str(k): [int(ncl[k - 1]), len(set(refs[k].tolist()) - {-1})] for k in refs},
                  "ari_vs_sklearn": {str(k): float(adjusted_rand_score(refs[k], labels[k - 1])) for k in refs}}))
