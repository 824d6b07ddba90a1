"""Ensemble sharding over the GPUs of one node: one process per GPU.

Ensemble members are independent (reference sampling.py:52-62 has no loop-carried state), so the path shards with
no data-path collective until the end: ONE variable-length gather of every rank's packed results.  Every rank
derives the same member -> rank table (`shard_items`, deterministic), hence the size of every rank's block: no
size exchange.  In the product the gather is `kbdm_plan_gather` of the C ABI (RCCL over xGMI, bound by the library
itself, device buffers to device buffers: `RcclComm`); the CPU tests run the same code over gloo (`GlooComm`).
A process launcher is only needed to start the ranks and to hand rank 0's communicator id to the others.
"""
import numpy as np

from . import _lib
from .kbdm import KbdmInfo, _resolve_m_l


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


def packed_bytes(lines, sv, members):
    """Size of a rank's packed block (include/kbdm_hip.h: kbdm_packed_bytes): lines x 4 f64 | sv f64 | status i32 |
    keep u8, padded to 16 bytes."""
    raw = 32 * int(lines) + 8 * int(sv) + 4 * int(members) + int(lines)
    return (raw + 15) & ~15


def pack_block(lines, sv, status, keep):
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
    return out


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
    (rank 0 passes the id it created, the others pass None).  Nothing else crosses Python."""

    def __init__(self, engine, world, rank, exchange_id, force=False):
        self.engine, self.world, self.rank = engine, int(world), int(rank)
        lib = engine.lib
        self.owns = self.world > 1 or force         # force: a one-rank communicator (rehearsal of the RCCL path)
        if self.owns:
            uid = None
            if self.rank == 0:
                buf = np.zeros(_lib.KBDM_UNIQUE_ID_BYTES, dtype=np.uint8)
                _lib.check(lib.kbdm_comm_unique_id(_lib.ptr(buf)))
                uid = buf.tobytes()
            uid = exchange_id(uid)
            buf = np.frombuffer(uid, dtype=np.uint8).copy()
            _lib.check(lib.kbdm_comm_init(engine.ctx, self.world, self.rank, _lib.ptr(buf)))

    def solve_and_gather(self, signals, ms, ls, p, q, dwell, sizes, root=-1):
        """Solve this rank's members and gather every rank's packed block (`sizes[r]` bytes each) on the device;
        returns the concatenated host copy on receiving ranks (None elsewhere)."""
        eng = self.engine
        n = len(ms)
        plan = eng.cached_plan(signals.shape[0], signals.shape[1], np.zeros(n, dtype=np.int32), ms, ls, p, q, dwell)
        plan.upload(signals)
        plan.execute(sync=False)
        return self.gather_plan(plan, sizes, root)

    def gather_plan(self, plan, sizes, root=-1):
        sizes = np.ascontiguousarray(sizes, dtype=np.int64)
        recv = root < 0 or root == self.rank
        out = np.empty(int(sizes.sum()), dtype=np.uint8) if recv else None
        _lib.check(self.engine.lib.kbdm_plan_gather(plan.handle, self.world, self.rank, _lib.ptr(sizes), int(root),
                                                    _lib.ptr(out)))
        return out

    def close(self):
        if self.owns:
            self.engine.lib.kbdm_comm_destroy(self.engine.ctx)
            self.owns = False


class GlooComm:
    """CPU stand-in with the same interface (tests): the per-rank solver is injected, the blocks are packed on the
    host in the library's layout and all-gathered over a torch.distributed (gloo) group."""

    def __init__(self, solve, group=None):
        import torch.distributed as dist
        self.solve, self.group = solve, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)

    def solve_and_gather(self, signals, ms, ls, p, q, dwell, sizes, root=-1):
        import torch
        import torch.distributed as dist
        n = len(ms)
        if n:
            res = self.solve(signals, np.zeros(n, dtype=np.int32), list(ms), list(ls), p=p, q=q, dwell=dwell)
            lines = np.concatenate([np.asarray(res.line_list(k)).reshape(-1, 4) for k in range(n)])
            keep = np.concatenate([np.asarray(res.keep_mask(k)) for k in range(n)])
            svs = np.concatenate([np.asarray(res.singular_values(k)) for k in range(n)])
            block = pack_block(lines, svs, np.zeros(n, np.int32), keep)
        else:
            block = pack_block(np.zeros((0, 4)), np.zeros(0), np.zeros(0, np.int32), np.zeros(0, np.uint8))
        assert block.size == sizes[self.rank], (block.size, sizes[self.rank])
        pad = int(max(max(sizes), 16))
        mine = torch.zeros(pad, dtype=torch.uint8)
        mine[:block.size] = torch.from_numpy(block)
        parts = [torch.zeros(pad, dtype=torch.uint8) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        return np.concatenate([parts[r][:int(sizes[r])].numpy() for r in range(self.world)])

    def close(self):
        pass


def sample_kbdm_sharded(data, dwell, m_range, p, l, q=0, filter_invalid_features=True, comm=None, root=-1):
    """Distributed drop-in for ``sample_kbdm`` (reference sampling.py:8-72): every rank solves its share of
    ``m_range`` and the receiving ranks (all of them unless ``root`` names one) get the complete
    ``(line_lists, infos)`` in ``m_range`` order; the other ranks get ``(None, None)``.

    ``comm``: an `RcclComm` (product) or `GlooComm` (CPU tests)."""
    if comm is None:
        raise ValueError("sample_kbdm_sharded needs a communicator (RcclComm / GlooComm)")
    data = np.asarray(data)
    ms, ls = [], []
    for m in m_range:
        mm, ll = _resolve_m_l(data.size, m, p, l)
        ms.append(mm)
        ls.append(ll)
    ms, ls = np.asarray(ms, dtype=np.int32), np.asarray(ls, dtype=np.int32)
    world, rank = comm.world, comm.rank
    parts = shard_items(ms.astype(np.float64) ** 3, world)
    sizes = np.array([packed_bytes(ls[idx].sum(), ms[idx].sum(), len(idx)) for idx in parts], dtype=np.int64)
    mine = parts[rank]
    buf = comm.solve_and_gather(np.ascontiguousarray(data, dtype=np.complex128).reshape(1, -1), ms[mine], ls[mine],
                                p, q, dwell, sizes, root)
    if buf is None:
        return None, None
    return unpack_gathered(buf, parts, sizes, ms, ls, p, q, filter_invalid_features)


def unpack_gathered(buf, parts, sizes, ms, ls, p, q, filter_invalid_features=True):
    """Gathered blocks (rank order) -> (line_lists, infos) in member order, empty results dropped
    (reference sampling.py:64-70)."""
    line_lists, infos = [None] * len(ms), [None] * len(ms)
    off = 0
    for r, idx in enumerate(parts):
        nl, nsv = int(ls[idx].sum()), int(ms[idx].sum())
        rl, rs, _, rk = unpack_block(buf[off:off + int(sizes[r])], nl, nsv, len(idx))
        off += int(sizes[r])
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
    out_l, out_i = [], []
    for ll_i, info in zip(line_lists, infos):
        if len(ll_i) > 0:                      # reference sampling.py:67-70
            out_l.append(ll_i)
            out_i.append(info)
    return out_l, out_i
