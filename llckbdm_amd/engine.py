"""Batch planner: turns (signals, items) into one call of the HIP pipeline.

This is the host half of the batched replacement for the reference's serial loop
``for m in m_range: kbdm(...)`` (llckbdm/sampling.py:52-70).
"""
import os
import threading
from collections import OrderedDict

import numpy as np

from . import _lib

_default = None
_default_lock = threading.Lock()


class BatchResult:
    """Flat outputs of one batch plus per-item views."""

    def __init__(self, lines, sv, mu, keep, status, line_off, sv_off):
        self.lines, self.sv, self.mu, self.keep, self.status = lines, sv, mu, keep, status
        self.line_off, self.sv_off = line_off, sv_off

    def __len__(self):
        return len(self.status)

    def line_list(self, i):
        return self.lines[self.line_off[i]:self.line_off[i + 1]]

    def keep_mask(self, i):
        return self.keep[self.line_off[i]:self.line_off[i + 1]].astype(bool)

    def singular_values(self, i):
        return self.sv[self.sv_off[i]:self.sv_off[i + 1]]

    def eigenvalues(self, i):
        return self.mu[self.line_off[i]:self.line_off[i + 1]]


class Plan:
    """A fixed batch geometry with its device workspace (kbdm_plan in include/kbdm_hip.h)."""

    def __init__(self, engine, S, N, sig_idx, m, l, p, q, dwell):
        self.engine = engine
        lib = engine.lib
        self.S, self.N, self.B = int(S), int(N), len(m)
        self.sig_idx = np.ascontiguousarray(sig_idx, dtype=np.int32)
        self.m = np.ascontiguousarray(m, dtype=np.int32)
        self.l = np.ascontiguousarray(l, dtype=np.int32)
        self.p, self.q, self.dwell = int(p), float(q), float(dwell)
        h = _lib.c_void_p()
        _lib.check(lib.kbdm_plan_create(engine.ctx, self.S, self.N, self.B, _lib.ptr(self.sig_idx),
                                        _lib.ptr(self.m), _lib.ptr(self.l), self.p, self.q, self.dwell, h))
        self.handle = h
        self.line_off = np.zeros(self.B + 1, dtype=np.int64)
        self.sv_off = np.zeros(self.B + 1, dtype=np.int64)
        _lib.check(lib.kbdm_plan_offsets(h, _lib.ptr(self.line_off), _lib.ptr(self.sv_off)))
        self.total_lines = int(lib.kbdm_plan_total_lines(h))
        self.total_sv = int(lib.kbdm_plan_total_sv(h))

    def upload(self, signals):
        sig = np.ascontiguousarray(signals, dtype=np.complex128).reshape(self.S, self.N)
        _lib.check(self.engine.lib.kbdm_plan_upload(self.handle, _lib.ptr(sig)))

    def execute(self, sync=True):
        _lib.check(self.engine.lib.kbdm_plan_execute(self.handle))
        if sync:
            self.sync()

    def sync(self):
        _lib.check(self.engine.lib.kbdm_plan_sync(self.handle))

    def wait_stage(self, name):
        """Block until the critical lane of the run in flight has finished stage ``name`` (e.g. ``"k_hess"``)."""
        lib = self.engine.lib
        names = [lib.kbdm_stage_name(i).decode() for i in range(_lib.KBDM_NSTAGES)]
        _lib.check(lib.kbdm_plan_wait_stage(self.handle, names.index(name)))

    def download(self):
        lines = np.empty((self.total_lines, 4), dtype=np.float64)
        sv = np.empty(self.total_sv, dtype=np.float64)
        mu = np.empty(self.total_lines, dtype=np.complex128)
        keep = np.empty(self.total_lines, dtype=np.uint8)
        status = np.empty(self.B, dtype=np.int32)
        _lib.check(self.engine.lib.kbdm_plan_download(self.handle, _lib.ptr(lines), _lib.ptr(sv), _lib.ptr(mu),
                                                      _lib.ptr(keep), _lib.ptr(status)))
        return BatchResult(lines, sv, mu, keep, status, self.line_off, self.sv_off)

    def stage_ms(self):
        ms = np.zeros(_lib.KBDM_NSTAGES, dtype=np.float32)
        _lib.check(self.engine.lib.kbdm_plan_stage_ms(self.handle, _lib.ptr(ms), _lib.KBDM_NSTAGES))
        names = [self.engine.lib.kbdm_stage_name(i).decode() for i in range(_lib.KBDM_NSTAGES)]
        return dict(zip(names, ms.tolist()))

    def lane0_members(self):
        """Members (the largest ones) in lane 0, whose stage timers `stage_ms` reports."""
        return int(self.engine.lib.kbdm_plan_lane0_members(self.handle))

    def lines_device_ptr(self):
        return self.engine.lib.kbdm_plan_lines_device(self.handle)

    def copy_lines_to_device(self, dst_ptr, dst_bytes):
        """D2D copy of the packed (total_lines, 4) float64 lines into a caller-owned device buffer."""
        _lib.check(self.engine.lib.kbdm_plan_copy_lines_device(self.handle, _lib.c_void_p(dst_ptr), int(dst_bytes)))

    def close(self):
        if self.handle is not None:
            self.engine.lib.kbdm_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """One GPU context (kbdm_ctx).  One per process and device."""

    def __init__(self, device=None):
        self.lib = _lib.load()
        if device is None:
            device = int(os.environ.get("KBDM_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        n = self.lib.kbdm_device_count()
        if n <= 0:
            raise _lib.KbdmHipError("no HIP device visible: llckbdm_amd needs an MI355X (gfx950); "
                                    "there is no CPU fallback")
        self.device = int(device) % n
        h = _lib.c_void_p()
        _lib.check(self.lib.kbdm_ctx_create(self.device, h))
        self.ctx = h
        # plans of recent `solve` calls, keyed on the batch geometry: a plan owns its device workspace (about 1 GB
        # for a C2 ensemble), and callers such as `sample_kbdm` / `iterative_llc_kbdm` solve the same geometry
        # again and again with new signals
        self._plan_cache = OrderedDict()
        self.plan_cache_size = int(os.environ.get("KBDM_PLAN_CACHE", "4"))

    def plan(self, S, N, sig_idx, m, l, p=1, q=0.0, dwell=1.0):
        return Plan(self, S, N, sig_idx, m, l, p, q, dwell)

    def cached_plan(self, S, N, sig_idx, m, l, p=1, q=0.0, dwell=1.0):
        """The plan for this batch geometry, created on first use and kept (LRU, `plan_cache_size` entries)."""
        sig_idx = np.ascontiguousarray(sig_idx, dtype=np.int32)
        m = np.ascontiguousarray(m, dtype=np.int32)
        l = np.ascontiguousarray(l, dtype=np.int32)
        key = (int(S), int(N), sig_idx.tobytes(), m.tobytes(), l.tobytes(), int(p), float(q), float(dwell))
        plan = self._plan_cache.get(key)
        if plan is not None and plan.handle is not None:
            self._plan_cache.move_to_end(key)
            return plan
        plan = self.plan(S, N, sig_idx, m, l, p, q, dwell)
        self._plan_cache[key] = plan
        while len(self._plan_cache) > max(1, self.plan_cache_size):
            _, old = self._plan_cache.popitem(last=False)
            old.close()
        return plan

    def clear_plan_cache(self):
        for plan in self._plan_cache.values():
            plan.close()
        self._plan_cache.clear()

    def solve(self, signals, sig_idx, m, l=None, p=1, q=0.0, dwell=1.0):
        """signals: (S, N) complex; items (sig_idx[i], m[i], l[i]).  Returns BatchResult.
        The plan (device workspace) of a geometry is reused by later calls with the same geometry."""
        signals = np.ascontiguousarray(np.atleast_2d(signals), dtype=np.complex128)
        m = np.asarray(m, dtype=np.int32)
        l = m.copy() if l is None else np.asarray(l, dtype=np.int32)
        plan = self.cached_plan(signals.shape[0], signals.shape[1], sig_idx, m, l, p, q, dwell)
        plan.upload(signals)
        plan.execute()
        return plan.download()

    # ---- stage entry points (parity tests) -------------------------------------------
    def hankel(self, signals, sig_idx, m, p):
        signals = np.ascontiguousarray(np.atleast_2d(signals), dtype=np.complex128)
        m = np.ascontiguousarray(m, dtype=np.int32)
        sig_idx = np.ascontiguousarray(sig_idx, dtype=np.int32)
        tot = int(np.sum(m.astype(np.int64) ** 2))
        outs = [np.empty(tot, dtype=np.complex128) for _ in range(3)]
        _lib.check(self.lib.kbdm_hankel_batch(self.ctx, _lib.ptr(signals), signals.shape[0], signals.shape[1],
                                              len(m), _lib.ptr(sig_idx), _lib.ptr(m), int(p),
                                              _lib.ptr(outs[0]), _lib.ptr(outs[1]), _lib.ptr(outs[2])))
        res, o = [], 0
        for mi in m:
            res.append(tuple(x[o:o + mi * mi].reshape(mi, mi) for x in outs))
            o += int(mi) * int(mi)
        return res

    def svd(self, mats):
        m = np.array([a.shape[0] for a in mats], dtype=np.int32)
        flat = np.concatenate([np.ascontiguousarray(a, dtype=np.complex128).ravel() for a in mats])
        L, R = np.empty_like(flat), np.empty_like(flat)
        s = np.empty(int(m.sum()), dtype=np.float64)
        status = np.zeros(len(m), dtype=np.int32)
        _lib.check(self.lib.kbdm_svd_batch(self.ctx, _lib.ptr(flat), len(m), _lib.ptr(m), _lib.ptr(L), _lib.ptr(s),
                                           _lib.ptr(R), _lib.ptr(status)))
        out, o, so = [], 0, 0
        for mi in m:
            mi = int(mi)
            out.append((L[o:o + mi * mi].reshape(mi, mi), s[so:so + mi], R[o:o + mi * mi].reshape(mi, mi)))
            o += mi * mi
            so += mi
        return out, status

    def rmse_batch(self, data, dwell, candidates):
        """Frequency-domain RMSE (reference metrics.py:7-17) of every candidate line list against `data`;
        an empty candidate scores +inf (min_rmse_kbdm.py:36-37).  One GPU call for all candidates."""
        data = np.ascontiguousarray(data, dtype=np.complex128).ravel()
        cands = [np.ascontiguousarray(np.asarray(c, dtype=np.float64).reshape(-1, 4)) for c in candidates]
        off = np.zeros(len(cands) + 1, dtype=np.int64)
        for i, c in enumerate(cands):
            off[i + 1] = off[i] + len(c)
        lines = np.concatenate(cands) if cands and off[-1] > 0 else np.zeros((1, 4))
        lines = np.ascontiguousarray(lines, dtype=np.float64)
        out = np.empty(len(cands), dtype=np.float64)
        _lib.check(self.lib.kbdm_rmse_batch(self.ctx, _lib.ptr(data), int(data.size), float(dwell), _lib.ptr(lines),
                                            _lib.ptr(off), len(cands), _lib.ptr(out)))
        return out

    def silhouette_samples(self, X, labels):
        """Silhouette coefficient of every sample, Euclidean metric (sklearn.metrics.silhouette_samples as used
        at llckbdm.py:291), computed on the GPU with directly evaluated distances."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        labels = np.ascontiguousarray(labels, dtype=np.int32)
        if X.ndim != 2 or labels.shape != (X.shape[0],):
            raise ValueError("X must be (n, dim) and labels (n,)")
        out = np.empty(X.shape[0], dtype=np.float64)
        _lib.check(self.lib.kbdm_silhouette_samples(self.ctx, _lib.ptr(X), X.shape[0], X.shape[1], _lib.ptr(labels),
                                                    _lib.ptr(out)))
        return out

    def hdbscan_sweep(self, X, min_samples_list, min_cluster_size=5):
        """HDBSCAN* labels for every `min_samples` of the list in one GPU call (the clustering sweep of
        llckbdm.py:104-110): one pass of k-nearest-neighbour distances shared by all fits, one Prim MST per fit
        (all fits concurrently), condensed trees on host threads.  Returns (labels[nfits, n], nclusters[nfits])."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        ms = np.ascontiguousarray(min_samples_list, dtype=np.int32)
        labels = np.empty((len(ms), X.shape[0]), dtype=np.int32)
        ncl = np.empty(len(ms), dtype=np.int32)
        _lib.check(self.lib.kbdm_hdbscan_sweep(self.ctx, _lib.ptr(X), X.shape[0], X.shape[1], _lib.ptr(ms), len(ms),
                                               int(min_cluster_size), _lib.ptr(labels), _lib.ptr(ncl)))
        return labels, ncl

    def eig(self, mats):
        n = np.array([a.shape[0] for a in mats], dtype=np.int32)
        flat = np.concatenate([np.ascontiguousarray(a, dtype=np.complex128).ravel() for a in mats])
        P = np.empty_like(flat)
        mu = np.empty(int(n.sum()), dtype=np.complex128)
        status = np.zeros(len(n), dtype=np.int32)
        _lib.check(self.lib.kbdm_eig_batch(self.ctx, _lib.ptr(flat), len(n), _lib.ptr(n), _lib.ptr(mu), _lib.ptr(P),
                                           _lib.ptr(status)))
        out, o, so = [], 0, 0
        for ni in n:
            ni = int(ni)
            out.append((mu[so:so + ni], P[o:o + ni * ni].reshape(ni, ni)))
            o += ni * ni
            so += ni
        return out, status

    def close(self):
        if self.ctx is not None:
            self.clear_plan_cache()
            self.lib.kbdm_ctx_destroy(self.ctx)
            self.ctx = None


def default_engine():
    """Process-wide engine on device LOCAL_RANK / KBDM_DEVICE (created on first use)."""
    global _default
    with _default_lock:
        if _default is None:
            _default = Engine()
        return _default
