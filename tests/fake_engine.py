"""CPU stand-in for llckbdm_amd.engine.Engine, for the `-m "not gpu"` tests of HOST LOGIC only (control flow of
llc_kbdm / min_rmse_kbdm, packing, error behaviour).  Every numerical entry point is answered by the oracle
(test infrastructure); nothing in the product can reach this class."""
import numpy as np

from llckbdm_amd.engine import BatchResult
from oracle import kbdm_oracle as O
from oracle import llc_oracle as L


class OracleEngine:
    def solve(self, signals, sig_idx, m, l=None, p=1, q=0.0, dwell=1.0, check=False):
        signals = np.atleast_2d(np.asarray(signals, dtype=np.complex128))
        m = np.asarray(m, dtype=np.int32)
        l = m.copy() if l is None else np.asarray(l, dtype=np.int32)
        lines, svs, keeps = [], [], []
        line_off, sv_off = [0], [0]
        for i, (mm, ll) in enumerate(zip(m, l)):
            line_list, info = O.kbdm(signals[sig_idx[i]], dwell, m=int(mm), p=p, l=int(ll), q=q, normalizer="gemm")
            lines.append(line_list)
            svs.append(np.asarray(info.singular_values))
            keeps.append(((line_list[:, 0] > 1e-6) & (line_list[:, 1] > 0)).astype(np.uint8))
            line_off.append(line_off[-1] + len(line_list))
            sv_off.append(sv_off[-1] + int(mm))
        return BatchResult(np.concatenate(lines), np.concatenate(svs), None, np.concatenate(keeps),
                           np.zeros(len(m), np.int32), np.array(line_off), np.array(sv_off))

    def rmse_batch(self, data, dwell, candidates):
        return np.array([L.calculate_freq_domain_rmse(data, c, dwell) if len(c) > 0 else np.inf for c in candidates])

    def silhouette_samples(self, X, labels):
        return L.silhouette_samples_direct(X, labels)

    def hdbscan_sweep(self, X, min_samples_list, min_cluster_size=5):
        from sklearn.cluster import HDBSCAN
        labels = np.array([HDBSCAN(min_samples=int(k), min_cluster_size=min_cluster_size, copy=True).fit(X).labels_
                           for k in min_samples_list], dtype=np.int32)
        return labels, np.array([len(set(row.tolist()) - {-1}) for row in labels], dtype=np.int32)
