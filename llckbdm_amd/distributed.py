"""Ensemble sharding over the GPUs of one node: one process per GPU.

Ensemble members are independent (reference sampling.py:52-62 has no loop-carried state), so the path shards with
no data-path collective until the end: ONE variable-length gather of every rank's packed results.  Every rank
derives the same member -> rank table (`shard_items`, deterministic), hence the size of every rank's block: no
size exchange.  In the product the gather is `kbdm_plan_gather` of the C ABI (RCCL over xGMI, bound by the library
itself, device buffers to device buffers: `RcclComm`); the CPU tests run the same sharding / packing / unpacking code
with the transport swapped (`HostComm` over `launch.Rendezvous`, standard library only).  A process launcher is only
needed to start the ranks and to hand rank 0's communicator id to the others (`llckbdm_amd.launch`).
"""
import warnings

import numpy as np

from . import _lib
from .kbdm import KbdmInfo, _check_finite, _resolve_m_l


def shard_items(costs, world_size):
    """Longest-processing-time assignment of items to ranks; returns one index array per rank.

    ``costs`` ~ m**3.  Deterministic (stable sort, lowest rank wins ties) so
    that every rank derives the same partition without communicating."""
    costs = np.asarray(costs, dtype=np.float64)
    order = np.argsort(-costs, kind="stable")
    load = np.zeros(world_size)
    buckets = [[] for _ in range(world_size)]
    for i in order:
        r = int(np.argmin(load))
        buckets[r].append(int(i))
        load[r] += costs[i]
    return [np.array(sorted(b), dtype=np.int64) for b in buckets]


TRAILER_BYTES = 16
TRAILER_MAGIC = 0x4B42444D      # "KBDM"


def packed_bytes(lines, sv, members):
    """Size of a rank's packed block (include/kbdm_hip.h: kbdm_packed_bytes): lines x 4 f64 | sv f64 | status i32 |
    keep u8, padded to 16 bytes, + the 16-byte trailer {u32 magic, u32 rank, u64 gather sequence number}."""
    raw = 32 * int(lines) + 8 * int(sv) + 4 * int(members) + int(lines)
    return ((raw + 15) & ~15) + TRAILER_BYTES


def pack_block(lines, sv, status, keep, rank=0, seq=0):
    """Host-side packing in the library's layout (used by the gloo stand-in and by tests)."""
    lines = np.ascontiguousarray(lines, dtype=np.float64).reshape(-1, 4)
    sv = np.ascontiguousarray(sv, dtype=np.float64).ravel()
    status = np.ascontiguousarray(status, dtype=np.int32).ravel()
    keep = np.ascontiguousarray(keep, dtype=np.uint8).ravel()
    out = np.zeros(packed_bytes(len(lines), len(sv), len(status)), dtype=np.uint8)
    o = 0
    for part in (lines, sv, status, keep):
        b = part.view(np.uint8).ravel()
        out[o:o + b.size] = b
        o += b.size
    out[-TRAILER_BYTES:] = np.frombuffer(np.array([TRAILER_MAGIC, int(rank)], dtype=np.uint32).tobytes() +
                                         np.array([int(seq)], dtype=np.uint64).tobytes(), dtype=np.uint8)
    return out


def block_trailer(buf):
    """(magic, rank, seq) of a packed block."""
    t = np.ascontiguousarray(buf, dtype=np.uint8)[-TRAILER_BYTES:]
    magic, rank = np.frombuffer(t[:8].tobytes(), dtype=np.uint32)
    return int(magic), int(rank), int(np.frombuffer(t[8:].tobytes(), dtype=np.uint64)[0])


def unpack_block(buf, lines, sv, members):
    """(lines[L,4] f64, sv[SV] f64, status[B] i32, keep[L] bool) views of one packed block."""
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    o = 0
    ll = buf[o:o + 32 * lines].view(np.float64).reshape(lines, 4)
    o += 32 * lines
    s = buf[o:o + 8 * sv].view(np.float64)
    o += 8 * sv
    st = buf[o:o + 4 * members].view(np.int32)
    o += 4 * members
    kp = buf[o:o + lines].astype(bool)
    return ll, s, st, kp


class RcclComm:
    """The product's communicator: RCCL inside libkbdm_hip.so, owned by the engine's context.

    ``exchange_id(id_or_None) -> id`` is the launcher's job: it must return rank 0's 128-byte id on every rank
    (rank 0 passes the id it created, the others pass None): `launch.Rendezvous.exchange_id`.  Nothing else crosses
    Python."""

    def __init__(self, engine, world, rank, exchange_id, force=False, ctx=None, share=None):
        self.engine, self.world, self.rank = engine, int(world), int(rank)
        self.ctx = engine.ctx if ctx is None else ctx
        lib = engine.lib
        self.owns = self.world > 1 or force         # force: a one-rank communicator (rehearsal of the RCCL path)
        if self.owns and share is not None:
            # another context of this process on the same GPU: borrow `share`'s communicator (kbdm_comm_attach)
            _lib.check(lib.kbdm_comm_attach(self.ctx, share.ctx))
        elif self.owns:
            uid = None
            if self.rank == 0:
                buf = np.zeros(_lib.KBDM_UNIQUE_ID_BYTES, dtype=np.uint8)
                _lib.check(lib.kbdm_comm_unique_id(_lib.ptr(buf)))
                uid = buf.tobytes()
            uid = exchange_id(uid)
            buf = np.frombuffer(uid, dtype=np.uint8).copy()
            _lib.check(lib.kbdm_comm_init(self.ctx, self.world, self.rank, _lib.ptr(buf)))

    def solve_and_gather(self, signals, sig_idx, ms, ls, p, q, dwell, sizes, root=-1):
        """Solve this rank's members and gather every rank's packed block (`sizes[r]` bytes each) on the device;
        returns the concatenated host copy on receiving ranks (None elsewhere).  A rank whose members report
        non-convergence runs its share once more in the conservative modes before the gather (the same single retry
        as `engine.checked`), so the status words that travel are final."""
        eng = self.engine
        plan = eng.cached_plan(signals.shape[0], signals.shape[1], sig_idx, ms, ls, p, q, dwell)
        plan.upload(signals)
        plan.execute(sync=True)
        if len(ms):
            status = plan.download_status()
            if np.any(status & (_lib.STAT_SVD_NOCONV | _lib.STAT_EIG_NOCONV)):
                plan.set_mode(_lib.MODE_SOLO_QR)
                try:
                    plan.execute(sync=True)
                finally:
                    plan.set_mode(0)
        return self.gather_plan(plan, sizes, root)

    def gather_plan(self, plan, sizes, root=-1, host=True):
        """The one collective of the path.  host=True: returns the concatenated host copy on receiving ranks (the call
        waits); host=False: only enqueues the device-to-device transfer (`wait` completes it) and returns None."""
        sizes = np.ascontiguousarray(sizes, dtype=np.int64)
        recv = root < 0 or root == self.rank
        out = np.empty(int(sizes.sum()), dtype=np.uint8) if (recv and host) else None
        _lib.check(self.engine.lib.kbdm_plan_gather(plan.handle, self.world, self.rank, _lib.ptr(sizes), int(root),
                                                    _lib.ptr(out)))
        if not host or (not recv):
            if host:
                self.wait()
            return None
        return out

    def wait(self):
        """Block until this context's last gather has landed."""
        _lib.check(self.engine.lib.kbdm_gather_wait(self.ctx))

    def close(self):
        if self.owns:
            self.engine.lib.kbdm_comm_destroy(self.ctx)
            self.owns = False


class HostComm:
    """CPU stand-in with the same interface (tests): the per-rank solver is injected (`solve(signals, sig_idx, m, l,
    p=, q=, dwell=) -> BatchResult`), the blocks are packed on the host in the library's layout - status words
    included - and all-gathered through `launch.Rendezvous` (one TCP socket per rank, standard library only)."""

    def __init__(self, solve, rdzv):
        self.solve, self.rdzv = solve, rdzv
        self.world, self.rank = rdzv.world, rdzv.rank

    def solve_and_gather(self, signals, sig_idx, ms, ls, p, q, dwell, sizes, root=-1):
        n = len(ms)
        if n:
            res = self.solve(signals, np.asarray(sig_idx, dtype=np.int32), list(ms), list(ls), p=p, q=q, dwell=dwell)
            lines = np.concatenate([np.asarray(res.line_list(k)).reshape(-1, 4) for k in range(n)])
            keep = np.concatenate([np.asarray(res.keep_mask(k)) for k in range(n)])
            svs = np.concatenate([np.asarray(res.singular_values(k)) for k in range(n)])
            status = np.zeros(n, np.int32) if getattr(res, "status", None) is None else np.asarray(res.status, np.int32)
            block = pack_block(lines, svs, status, keep)
        else:
            block = pack_block(np.zeros((0, 4)), np.zeros(0), np.zeros(0, np.int32), np.zeros(0, np.uint8))
        assert block.size == sizes[self.rank], (block.size, sizes[self.rank])
        parts = self.rdzv.allgather(block.tobytes())
        if root >= 0 and root != self.rank:
            return None
        return np.concatenate([np.frombuffer(b, dtype=np.uint8) for b in parts])

    def close(self):
        pass


def sample_kbdm_signals_sharded(signals, dwell, sig_idx, m_list, p=1, l=None, q=0, filter_invalid_features=True,
                                comm=None, root=-1):
    """Distributed form of `sampling.sample_kbdm_signals` (the loop sampling.py:52-62 over a grid of signals, e.g.
    the 64 voxels x 256 members of a multi-voxel MRSI grid): item i = (signals[sig_idx[i]], m_list[i]).  The items
    are dealt over the ranks by LPT on m^3 (every rank derives the same table); every rank uploads only the signals
    its items use.  Receiving ranks (all unless `root` names one) get ``(line_lists, infos, item_index)`` of the
    non-empty members in item order; the others ``(None, None, None)``.  Members whose status word still reports
    non-convergence after the per-rank retry raise numpy.linalg.LinAlgError on every receiving rank.

    ``comm``: an `RcclComm` (product) or `HostComm` (CPU tests)."""
    if comm is None:
        raise ValueError("the sharded samplers need a communicator (RcclComm / HostComm)")
    signals = np.atleast_2d(np.asarray(signals))
    ms, ls = [], []
    for m in m_list:
        mm, ll = _resolve_m_l(signals.shape[1], m, p, l)
        ms.append(mm)
        ls.append(ll)
    ms, ls = np.asarray(ms, dtype=np.int32), np.asarray(ls, dtype=np.int32)
    sig_idx = np.asarray(sig_idx, dtype=np.int32)
    if len(ms):                                  # ValueError for NaN / Inf samples on EVERY rank, before anything is dealt
        wid = np.zeros(signals.shape[0], dtype=np.int64)
        np.maximum.at(wid, sig_idx, ms.astype(np.int64))
        for k in np.nonzero(wid)[0]:
            _check_finite(signals[k], int(wid[k]), p)
    world, rank = comm.world, comm.rank
    parts = shard_items(ms.astype(np.float64) ** 3, world)
    sizes = np.array([packed_bytes(ls[idx].sum(), ms[idx].sum(), len(idx)) for idx in parts], dtype=np.int64)
    mine = parts[rank]
    used = np.unique(sig_idx[mine]) if len(mine) else np.zeros(1, dtype=np.int32)
    remap = np.zeros(signals.shape[0], dtype=np.int32)
    remap[used] = np.arange(len(used), dtype=np.int32)
    buf = comm.solve_and_gather(np.ascontiguousarray(signals[used], dtype=np.complex128), remap[sig_idx[mine]],
                                ms[mine], ls[mine], p, q, dwell, sizes, root)
    if buf is None:
        return None, None, None
    return unpack_gathered(buf, parts, sizes, ms, ls, p, q, filter_invalid_features, with_index=True)


def sample_kbdm_sharded(data, dwell, m_range, p, l, q=0, filter_invalid_features=True, comm=None, root=-1):
    """Distributed drop-in for ``sample_kbdm`` (reference sampling.py:8-72): every rank solves its share of
    ``m_range`` and the receiving ranks (all of them unless ``root`` names one) get the complete
    ``(line_lists, infos)`` in ``m_range`` order; the other ranks get ``(None, None)``."""
    data = np.asarray(data)
    m_list = list(m_range)
    out = sample_kbdm_signals_sharded(data.reshape(1, -1), dwell, np.zeros(len(m_list), dtype=np.int32), m_list, p=p, l=l,
                                      q=q, filter_invalid_features=filter_invalid_features, comm=comm, root=root)
    return out[0], out[1]


def unpack_gathered(buf, parts, sizes, ms, ls, p, q, filter_invalid_features=True, with_index=False):
    """Gathered blocks (rank order) -> (line_lists, infos) in member order, empty results dropped
    (reference sampling.py:64-70).  The status words travel with the blocks: SVD_NOCONV / EIG_NOCONV raise
    numpy.linalg.LinAlgError (what scipy.linalg.svd / eig do inside the reference's kbdm(): kbdm.py:166,192),
    INVIT_WEAK warns."""
    from .engine import KbdmAccuracyWarning
    line_lists, infos = [None] * len(ms), [None] * len(ms)
    status = np.zeros(len(ms), dtype=np.int32)
    off = 0
    for r, idx in enumerate(parts):
        nl, nsv = int(ls[idx].sum()), int(ms[idx].sum())
        rl, rs, rst, rk = unpack_block(buf[off:off + int(sizes[r])], nl, nsv, len(idx))
        off += int(sizes[r])
        status[idx] = rst
        o = so = 0
        for i in idx:
            li, mi = int(ls[i]), int(ms[i])
            ll_i = rl[o:o + li]
            if filter_invalid_features:
                ll_i = ll_i[rk[o:o + li]]
            line_lists[i] = ll_i.copy()
            infos[i] = KbdmInfo(m=mi, l=li, p=p, q=q, singular_values=rs[so:so + mi].copy())
            o += li
            so += mi
    hard = np.nonzero(status & (_lib.STAT_SVD_NOCONV | _lib.STAT_EIG_NOCONV))[0]
    if len(hard):
        raise np.linalg.LinAlgError("sharded KBDM ensemble: SVD / eig did not converge for member(s) %s (m = %s, status %s)"
                                    % ([int(i) for i in hard], [int(ms[i]) for i in hard], [int(status[i]) for i in hard]))
    weak = np.nonzero(status & _lib.STAT_INVIT_WEAK)[0]
    if len(weak):
        warnings.warn(f"sharded KBDM ensemble: inverse iteration was weak for member(s) {[int(i) for i in weak]}",
                      KbdmAccuracyWarning, stacklevel=2)
    out_l, out_i, out_x = [], [], []
    for i, (ll_i, info) in enumerate(zip(line_lists, infos)):
        if len(ll_i) > 0:                      # reference sampling.py:67-70
            out_l.append(ll_i)
            out_i.append(info)
            out_x.append(i)
    return (out_l, out_i, out_x) if with_index else (out_l, out_i)
