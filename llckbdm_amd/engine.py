"""Batch planner: turns (signals, items) into one call of the HIP pipeline.

This is the host half of the batched replacement for the reference's serial loop
``for m in m_range: kbdm(...)`` (llckbdm/sampling.py:52-70).

`Engine.solve` runs one batch; `Engine.submit` / `Pending.result` keep up to `in_flight` independent batches on the
GPU at once (each on a context of its own: three HIP streams), which is how the throughput-bound stages of one
ensemble run underneath the latency-bound QR iterations of another.  Results carry the solver's per-member status
word; `checked` turns it into what the reference's LAPACK calls do on failure (scipy.linalg.svd / eig raise
LinAlgError: llckbdm/kbdm.py:166,192) after ONE conservative retry of the flagged members.
"""
import os
import threading
import warnings
from collections import OrderedDict

import numpy as np

from . import _lib

_default = None
_default_lock = threading.Lock()


class KbdmAccuracyWarning(UserWarning):
    """A member's eigenvectors came from an inverse iteration that did not reach its growth bound
    (status bit INVIT_WEAK): its lines may be less accurate than 1e-8."""


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


def checked(engine, res, signals, sig_idx, m, l, p, q, dwell, what="KBDM"):
    """The status word as the drop-in contract: members flagged SVD_NOCONV / EIG_NOCONV are solved ONCE more in the
    conservative mode (solo QR iteration: no workgroup teams; same process) and patched into `res`; members
    that are still flagged raise numpy.linalg.LinAlgError as scipy.linalg.svd / scipy.linalg.eig do in the reference
    (kbdm.py:166,192); INVIT_WEAK warns.  Returns `res`."""
    hard = _lib.STAT_SVD_NOCONV | _lib.STAT_EIG_NOCONV
    bad = np.nonzero(res.status & hard)[0]
    if len(bad):
        m = np.asarray(m, dtype=np.int32)
        l = np.asarray(l, dtype=np.int32)
        sig_idx = np.asarray(sig_idx, dtype=np.int32)
        signals = np.ascontiguousarray(np.atleast_2d(signals), dtype=np.complex128)
        plan = engine.plan(signals.shape[0], signals.shape[1], sig_idx[bad], m[bad], l[bad], p, q, dwell)
        try:
            plan.set_mode(_lib.MODE_SOLO_QR)
            plan.submit(signals)
            again = plan.collect()
        finally:
            plan.close()
        for k, i in enumerate(bad):
            res.lines[res.line_off[i]:res.line_off[i + 1]] = again.line_list(k)
            res.mu[res.line_off[i]:res.line_off[i + 1]] = again.eigenvalues(k)
            res.keep[res.line_off[i]:res.line_off[i + 1]] = again.keep[again.line_off[k]:again.line_off[k + 1]]
            res.sv[res.sv_off[i]:res.sv_off[i + 1]] = again.singular_values(k)
            res.status[i] = again.status[k]
        still = np.nonzero(res.status & hard)[0]
        if len(still) and q == 0:
            # an exactly singular U^{p-1} (a zero singular value among the l retained ones): the reference inverts
            # sqrt(diag(s)) there (kbdm.py:168-186) and numpy raises "Singular matrix"
            sing = [int(i) for i in still if not np.all(res.sv[res.sv_off[i]:res.sv_off[i] + int(l[i])] > 0)]
            if sing:
                raise np.linalg.LinAlgError("Singular matrix")
        if len(still):
            svd = [int(i) for i in still if res.status[i] & _lib.STAT_SVD_NOCONV]
            eig = [int(i) for i in still if res.status[i] & _lib.STAT_EIG_NOCONV]
            parts = []
            if svd:
                parts.append("SVD did not converge for member(s) %s (m = %s)" % (svd, [int(m[i]) for i in svd]))
            if eig:
                parts.append("eig algorithm did not converge for member(s) %s (l = %s)" % (eig, [int(l[i]) for i in eig]))
            raise np.linalg.LinAlgError(f"{what}: " + "; ".join(parts))
    weak = np.nonzero(res.status & _lib.STAT_INVIT_WEAK)[0]
    if len(weak):
        warnings.warn(f"{what}: inverse iteration was weak for member(s) {[int(i) for i in weak]}: their lines may be "
                      "less accurate than 1e-8", KbdmAccuracyWarning, stacklevel=3)
    return res


class Plan:
    """A fixed batch geometry with its device workspace (kbdm_plan in include/kbdm_hip.h)."""

    def __init__(self, engine, S, N, sig_idx, m, l, p, q, dwell, ctx=None):
        self.engine = engine
        lib = engine.lib
        ctx = engine.ctx if ctx is None else ctx
        self.S, self.N, self.B = int(S), int(N), len(m)
        self.sig_idx = np.ascontiguousarray(sig_idx, dtype=np.int32)
        self.m = np.ascontiguousarray(m, dtype=np.int32)
        self.l = np.ascontiguousarray(l, dtype=np.int32)
        self.p, self.q, self.dwell = int(p), float(q), float(dwell)
        h = _lib.c_void_p()
        _lib.check(lib.kbdm_plan_create(ctx, self.S, self.N, self.B, _lib.ptr(self.sig_idx),
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

    def set_mode(self, mode):
        """Conservative execution modes (`_lib.MODE_*`, include/kbdm_hip.h KBDM_MODE_*)."""
        _lib.check(self.engine.lib.kbdm_plan_set_mode(self.handle, int(mode)))

    def workspace_bytes(self):
        return int(self.engine.lib.kbdm_plan_workspace_bytes(self.handle))

    def submit(self, signals=None):
        """Host -> host without blocking: stage the signals (None = keep the uploaded ones), enqueue H2D, the
        pipeline and the D2H of every output; `collect` waits and returns the BatchResult."""
        sig = None
        if signals is not None:
            sig = np.ascontiguousarray(signals, dtype=np.complex128).reshape(self.S, self.N)
        _lib.check(self.engine.lib.kbdm_plan_submit(self.handle, _lib.ptr(sig)))

    def collect(self):
        lines = np.empty((self.total_lines, 4), dtype=np.float64)
        sv = np.empty(self.total_sv, dtype=np.float64)
        mu = np.empty(self.total_lines, dtype=np.complex128)
        keep = np.empty(self.total_lines, dtype=np.uint8)
        status = np.empty(self.B, dtype=np.int32)
        _lib.check(self.engine.lib.kbdm_plan_collect(self.handle, _lib.ptr(lines), _lib.ptr(sv), _lib.ptr(mu),
                                                     _lib.ptr(keep), _lib.ptr(status)))
        return BatchResult(lines, sv, mu, keep, status, self.line_off, self.sv_off)

    def download(self):
        lines = np.empty((self.total_lines, 4), dtype=np.float64)
        sv = np.empty(self.total_sv, dtype=np.float64)
        mu = np.empty(self.total_lines, dtype=np.complex128)
        keep = np.empty(self.total_lines, dtype=np.uint8)
        status = np.empty(self.B, dtype=np.int32)
        _lib.check(self.engine.lib.kbdm_plan_download(self.handle, _lib.ptr(lines), _lib.ptr(sv), _lib.ptr(mu),
                                                      _lib.ptr(keep), _lib.ptr(status)))
        return BatchResult(lines, sv, mu, keep, status, self.line_off, self.sv_off)

    def download_status(self):
        """The per-member status words alone (waits for the plan)."""
        status = np.empty(self.B, dtype=np.int32)
        _lib.check(self.engine.lib.kbdm_plan_download(self.handle, None, None, None, None, _lib.ptr(status)))
        return status

    def stage_ms(self):
        ms = np.zeros(_lib.KBDM_NSTAGES, dtype=np.float32)
        _lib.check(self.engine.lib.kbdm_plan_stage_ms(self.handle, _lib.ptr(ms), _lib.KBDM_NSTAGES))
        names = [self.engine.lib.kbdm_stage_name(i).decode() for i in range(_lib.KBDM_NSTAGES)]
        return dict(zip(names, ms.tolist()))

    def kernel_ms(self):
        """Per-kernel HIP-event timers of the last run in `_lib.MODE_KERNEL_TIMERS` mode (lane 0's launches):
        {kernel class: (total ms, launches)}."""
        lib = self.engine.lib
        out = {}
        for k in range(_lib.KBDM_NKCLASSES):
            ms = np.zeros(1, dtype=np.float32)
            n = np.zeros(1, dtype=np.int32)
            _lib.check(lib.kbdm_plan_kernel_ms(self.handle, k, _lib.ptr(ms), _lib.ptr(n)))
            out[lib.kbdm_kernel_class_name(k).decode()] = (float(ms[0]), int(n[0]))
        return out

    def lane0_members(self):
        """Members (the largest ones) in lane 0, whose stage timers `stage_ms` reports."""
        return int(self.engine.lib.kbdm_plan_lane0_members(self.handle))

    def eig_fallbacks(self):
        """Members of the last run that the Ehrlich-Aberth eigenvalue path handed to the QR iteration."""
        return int(self.engine.lib.kbdm_plan_eig_fallbacks(self.handle))

    def ab_stats(self):
        """Working root tiles per (step, iteration) of the Ehrlich-Aberth path since the last call: array (16, 24)."""
        out = np.zeros(16 * 24, dtype=np.int32)
        _lib.check(self.engine.lib.kbdm_plan_ab_stats(self.handle, _lib.ptr(out), out.size))
        return out.reshape(16, 24)

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


class Pending:
    """A batch in flight (`Engine.submit`)."""

    def __init__(self, engine, slot, plan, args):
        self._engine, self._slot, self._plan, self._args = engine, slot, plan, args
        self.plan = plan                   # stays valid (stage timers, device results) until its context runs again
        self._res = None

    def _finish(self):
        if self._res is None:
            self._res = self._plan.collect()
            self._slot.pending = None
            self._plan = None
        return self._res

    def wait(self):
        """Wait for the batch without copying anything out of the staging buffers yet."""
        self.plan.sync()

    def result(self, check=True):
        """Wait for the batch; `check` applies the status contract (`checked`: retry once, then LinAlgError)."""
        with self._engine._lock:
            res = self._finish()
            if check and self._args is not None:
                args, self._args = self._args, None
                checked(self._engine, res, *args)
            return res


class _Slot:
    """One context (kbdm_ctx: two lanes + a side stream) with the plans it has cached."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.plans = OrderedDict()
        self.pending = None
        self.seq = 0


class Engine:
    """One GPU.  Owns up to `in_flight` contexts (kbdm_ctx, created on demand) so that independent batches overlap."""

    # the second / third batch of a burst starts when the one before it has reached this stage: pipelines with the
    # same cycle keep the phase they start in, and started together they meet panels against panels
    STAGGER_STAGE = {2: "k_hess", 3: "k_dc_sv", 4: "k_gen(Q,P)"}

    def __init__(self, device=None, in_flight=None):
        self.lib = _lib.load()
        if device is None:
            device = int(os.environ.get("KBDM_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        n = self.lib.kbdm_device_count()
        if n <= 0:
            raise _lib.KbdmHipError("no HIP device visible: llckbdm_amd needs an MI355X (gfx950); "
                                    "there is no CPU fallback")
        self.device = int(device) % n
        self.in_flight = max(1, int(os.environ.get("KBDM_IN_FLIGHT", "4") if in_flight is None else in_flight))
        self._lock = threading.RLock()
        self._wide = None
        self.wide_solve = os.environ.get("KBDM_WIDE_SOLVE", "1") != "0"   # synchronous batches on the context with panel teams
        # The runtime hands out its hardware queues (GPU_MAX_HW_QUEUES, 16 asked for by the package) in order of stream
        # creation and streams that share one serialise: the wide context (three lanes + a side stream) comes first, then
        # the pool's contexts (three streams each) - a fifth context created late would run its lanes one after the other.
        if self.wide_solve:
            self._wide_slot()
        self._slots = [_Slot(self._new_ctx())]
        self.ctx = self._slots[0].ctx
        self._seq = 0
        self._burst = 0
        self._last = None
        self.stagger = os.environ.get("KBDM_STAGGER", "1") != "0"
        # Plans of recent batches, keyed on the batch geometry: a plan owns its device workspace (about 1 GB for a C2
        # ensemble, up to the 96 GiB workspace budget for a C4-sized one), and callers such as `sample_kbdm` /
        # `iterative_llc_kbdm` solve the same geometry again and again with new signals.  The cache is bounded by
        # BYTES over all contexts; idle plans are evicted (least recently used first) BEFORE a new one is allocated.
        self.plan_cache_bytes = int(float(os.environ.get("KBDM_PLAN_CACHE_GIB", "150")) * 2 ** 30)
        self.plan_cache_size = int(os.environ.get("KBDM_PLAN_CACHE", "4"))       # plans per context

    def _new_ctx(self):
        """A context of the in-flight pool (two lanes, no panel teams; KBDM_LANES / KBDM_PANEL_* are read by the library)."""
        h = _lib.c_void_p()
        _lib.check(self.lib.kbdm_ctx_create_lanes(self.device, 0, h))
        return h

    # Cooperative panels (include/kbdm_hip.h: kbdm_ctx_set_panel_teams).  The workgroups of a team wait for each other, so
    # the contexts that can run at the same time must not ask for more resident workgroups than the GPU has compute units:
    # the contexts of the in-flight pool run without teams (their panels overlap other ensembles' work instead); ONE
    # synchronous batch at a time - `solve`: kbdm, sample_kbdm, llc_kbdm - runs on a context of its own with teams of up
    # to WIDE_T workgroups for its largest members and three lanes.  Same bits either way.
    WIDE_T, WIDE_BUDGET, WIDE_LANES, WIDE_LANE0_FRAC = 4, 192, 3, 0.6

    def _wide_slot(self):
        if self._wide is None:
            h = _lib.c_void_p()
            _lib.check(self.lib.kbdm_ctx_create_lanes(self.device, int(os.environ.get("KBDM_WIDE_LANES", self.WIDE_LANES)), h))
            _lib.check(self.lib.kbdm_ctx_set_panel_teams(h, int(os.environ.get("KBDM_WIDE_T", self.WIDE_T)),
                                                         int(os.environ.get("KBDM_WIDE_BUDGET", self.WIDE_BUDGET)), 0,
                                                         float(os.environ.get("KBDM_WIDE_LANE0_FRAC", self.WIDE_LANE0_FRAC))))
            self._wide = _Slot(h)
        return self._wide

    def plan(self, S, N, sig_idx, m, l, p=1, q=0.0, dwell=1.0, ctx=None):
        return Plan(self, S, N, sig_idx, m, l, p, q, dwell, ctx=ctx)

    # ---- plan cache -------------------------------------------------------------------
    def _held_bytes(self):
        return sum(pl.workspace_bytes() for sl in self._all_slots() for pl in sl.plans.values() if pl.handle is not None)

    def _evict(self, need, keep_slot=None, everything=False):
        """Close idle cached plans, least recently used first, until `need` more bytes fit the budget."""
        cands = []
        for sl in self._all_slots():
            busy = sl.pending._plan if sl.pending is not None else None
            for key, pl in sl.plans.items():
                if pl is not busy:
                    cands.append((getattr(pl, "_used", 0), sl, key, pl))
        cands.sort(key=lambda c: c[0])
        held = self._held_bytes()
        for _, sl, key, pl in cands:
            if not everything and held + need <= self.plan_cache_bytes:
                break
            held -= pl.workspace_bytes()
            pl.close()
            del sl.plans[key]

    def cached_plan(self, S, N, sig_idx, m, l, p=1, q=0.0, dwell=1.0, slot=None):
        """The plan for this batch geometry on context `slot` (default: the first), created on first use and kept."""
        with self._lock:
            sl = self._slots[0] if slot is None else slot
            sig_idx = np.ascontiguousarray(sig_idx, dtype=np.int32)
            m = np.ascontiguousarray(m, dtype=np.int32)
            l = np.ascontiguousarray(l, dtype=np.int32)
            key = (int(S), int(N), sig_idx.tobytes(), m.tobytes(), l.tobytes(), int(p), float(q), float(dwell))
            plan = sl.plans.get(key)
            self._seq += 1
            if plan is not None and plan.handle is not None:
                sl.plans.move_to_end(key)
                plan._used = self._seq
                return plan
            est = int(self.lib.kbdm_workspace_estimate(len(m), _lib.ptr(m), _lib.ptr(l)))
            est = min(est, int(float(os.environ.get("KBDM_WS_GIB", "96")) * 2 ** 30))
            self._evict(est)
            while len(sl.plans) >= max(1, self.plan_cache_size):
                busy = sl.pending._plan if sl.pending is not None else None
                k0 = next((k for k, pl in sl.plans.items() if pl is not busy), None)
                if k0 is None:
                    break
                sl.plans.pop(k0).close()
            try:
                plan = self.plan(S, N, sig_idx, m, l, p, q, dwell, ctx=sl.ctx)
            except _lib.KbdmHipError:
                self._evict(0, everything=True)            # out of device memory: drop every idle plan, try once more
                plan = self.plan(S, N, sig_idx, m, l, p, q, dwell, ctx=sl.ctx)
            plan._used = self._seq
            sl.plans[key] = plan
            return plan

    def clear_plan_cache(self):
        with self._lock:
            self.drain()
            for sl in self._all_slots():
                for plan in sl.plans.values():
                    plan.close()
                sl.plans.clear()

    # ---- batches ----------------------------------------------------------------------
    def _pick_slot(self, key_fn):
        idle = [sl for sl in self._slots if sl.pending is None]
        if not idle and len(self._slots) < self.in_flight:
            self._slots.append(_Slot(self._new_ctx()))
            idle = [self._slots[-1]]
        if not idle:
            oldest = min(self._slots, key=lambda sl: sl.seq)
            oldest.pending._finish()
            idle = [oldest]
        for sl in idle:                       # an idle context that already holds this geometry's workspace
            if key_fn(sl):
                return sl
        return idle[0]

    def ensure_contexts(self, n=None):
        """Create the contexts of the pool up front (they are otherwise created when first needed)."""
        with self._lock:
            while len(self._slots) < (self.in_flight if n is None else min(n, self.in_flight)):
                self._slots.append(_Slot(self._new_ctx()))
            return [sl.ctx for sl in self._slots]

    def submit(self, signals, sig_idx, m, l=None, p=1, q=0.0, dwell=1.0, resident=False, _wide=False):
        """Start one batch (host signals in, host results out) and return at once: a `Pending`.  Up to `in_flight`
        batches run concurrently, each on its own context; a further submit first waits for the oldest one.
        `resident`: skip the upload when the plan that takes the batch already holds exactly this `signals` object
        (benchmarks with inputs resident in HBM)."""
        with self._lock:
            sig_obj = signals
            signals = np.ascontiguousarray(np.atleast_2d(signals), dtype=np.complex128)
            m = np.ascontiguousarray(m, dtype=np.int32)
            l = m.copy() if l is None else np.ascontiguousarray(l, dtype=np.int32)
            sig_idx = np.ascontiguousarray(sig_idx, dtype=np.int32)
            S, N = signals.shape
            key = (int(S), int(N), sig_idx.tobytes(), m.tobytes(), l.tobytes(), int(p), float(q), float(dwell))
            if all(sl.pending is None for sl in self._slots):
                self._burst = 0
            if self._wide is not None and self._wide.pending is not None:
                self._wide.pending._finish()              # the wide context never runs beside the pool (resident-team budget)
            if _wide and all(sl.pending is None for sl in self._slots):
                sl = self._wide_slot()
            else:
                sl = self._pick_slot(lambda s: key in s.plans and s.plans[key].handle is not None)
            plan = self.cached_plan(S, N, sig_idx, m, l, p, q, dwell, slot=sl)
            nfl = self.in_flight
            if self.stagger and 1 <= self._burst < nfl and self._last is not None and self._last._plan is not None \
                    and self._last._slot is not sl:
                try:
                    self._last._plan.wait_stage(self.STAGGER_STAGE.get(nfl, "k_svd_fac"))
                except ValueError:
                    pass
            if resident and getattr(plan, "_sig_obj", None) is sig_obj:
                plan.submit(None)
            else:
                plan.submit(signals)
                plan._sig_obj = sig_obj if resident else None
            self._seq += 1
            sl.seq = self._seq
            pend = Pending(self, sl, plan, (signals, sig_idx, m, l, p, q, dwell))
            sl.pending = pend
            self._last = pend
            self._burst += 1
            return pend

    def _all_slots(self):
        return self._slots + ([self._wide] if self._wide is not None else [])

    def drain(self):
        """Wait for every batch in flight (their results stay with their `Pending`)."""
        with self._lock:
            for sl in self._all_slots():
                if sl.pending is not None:
                    sl.pending._finish()

    def solve(self, signals, sig_idx, m, l=None, p=1, q=0.0, dwell=1.0, check=False):
        """signals: (S, N) complex; items (sig_idx[i], m[i], l[i]).  Returns BatchResult (raw status word unless
        `check`).  The plan (device workspace) of a geometry is reused by later calls with the same geometry."""
        return self.submit(signals, sig_idx, m, l, p, q, dwell, _wide=self.wide_solve).result(check=check)

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

    SWEEP_CHUNK_ENTRIES = 32_000_000

    def silhouette_sweep(self, X, labels):
        """`silhouette_samples` for every row of `labels` (nfits x n) in one GPU call: (silhouettes[nfits, n], valid[nfits]);
        the same bits as one call per row; a row outside sklearn's precondition comes back with valid = False."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        labels = np.ascontiguousarray(np.atleast_2d(labels), dtype=np.int32)
        out = np.empty(labels.shape, dtype=np.float64)
        valid = np.empty(labels.shape[0], dtype=np.int32)
        # a call holds two index arrays and the silhouettes of its fits on the host and on the device: at most 32 M entries each
        step = max(1, self.SWEEP_CHUNK_ENTRIES // max(1, X.shape[0]))
        for f0 in range(0, labels.shape[0], step):
            lab, o, v = labels[f0:f0 + step], out[f0:f0 + step], valid[f0:f0 + step]
            _lib.check(self.lib.kbdm_silhouette_sweep(self.ctx, _lib.ptr(X), X.shape[0], X.shape[1], _lib.ptr(lab), lab.shape[0],
                                                      _lib.ptr(o), _lib.ptr(v)))
        return out, valid.astype(bool)

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

    def core_distances(self, X, min_samples_list):
        """Distance of every sample to its k-th nearest sample (itself counted) for every k of the list: [nfits, n].
        The k-nearest-neighbour pass of `hdbscan_sweep` on its own."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        ms = np.ascontiguousarray(min_samples_list, dtype=np.int32)
        out = np.empty((len(ms), X.shape[0]), dtype=np.float64)
        _lib.check(self.lib.kbdm_core_distances(self.ctx, _lib.ptr(X), X.shape[0], X.shape[1], _lib.ptr(ms), len(ms), _lib.ptr(out)))
        return out

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

    def last_eig_fallbacks(self):
        """Members the Ehrlich-Aberth path handed to the QR iteration in the last `eig` call of this engine."""
        return int(self.lib.kbdm_ctx_last_eig_fallbacks(self.ctx))

    def close(self):
        if self.ctx is not None:
            self.clear_plan_cache()
            for sl in self._all_slots():
                self.lib.kbdm_ctx_destroy(sl.ctx)
            self._slots = []
            self._wide = None
            self.ctx = None


    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def default_engine():
    """Process-wide engine on device LOCAL_RANK / KBDM_DEVICE (created on first use)."""
    global _default
    with _default_lock:
        if _default is None:
            _default = Engine()
        return _default
