"""Ensemble sharding over the GPUs of one node (one process per GPU, torch.distributed).

Ensemble members are independent (reference sampling.py:52-62 has no loop-carried state), so
the path shards with no data-path collective until the end: ONE variable-length gather of the
packed line lists (+ singular values) - RCCL over xGMI when the process group is "nccl", gloo
in the CPU tests.  torch is plumbing here (process group, device buffers); it is imported
lazily and only by this module.
"""
import numpy as np

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


def _gather_varlen(local, dist, group, device):
    """All-gather of 1-D float64 arrays of different lengths: sizes first, then padded payloads."""
    import torch
    world = dist.get_world_size(group)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    pad = max(sizes) if sizes else 0
    buf = torch.zeros(max(pad, 1), dtype=torch.float64, device=device)
    if local.shape[0]:
        buf[:local.shape[0]] = torch.from_numpy(np.ascontiguousarray(local)).to(device)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return [o[:s].cpu().numpy() for o, s in zip(out, sizes)]


def sample_kbdm_sharded(data, dwell, m_range, p, l, q=0, filter_invalid_features=True, group=None,
                        solve=None, device=None):
    """Distributed drop-in for ``sample_kbdm``: every rank solves its share of ``m_range`` and all
    ranks receive the complete ``(line_lists, infos)`` in ``m_range`` order.

    ``solve(signals, sig_idx, ms, ls, p, q, dwell)`` must return an object with the BatchResult
    interface; by default it is the HIP engine of this rank's GPU.
    """
    import torch
    import torch.distributed as dist
    data = np.asarray(data)
    ms, ls = [], []
    for m in m_range:
        mm, ll = _resolve_m_l(data.size, m, p, l)
        ms.append(mm)
        ls.append(ll)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    parts = shard_items(np.asarray(ms, dtype=np.float64) ** 3, world)
    mine = parts[rank]
    if solve is None:
        from .engine import default_engine
        solve = default_engine().solve
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" \
            else torch.device("cpu")
    if len(mine):
        res = solve(data.reshape(1, -1), np.zeros(len(mine), dtype=np.int32), [ms[i] for i in mine],
                    [ls[i] for i in mine], p=p, q=q, dwell=dwell)
        lines = np.concatenate([res.line_list(k).ravel() for k in range(len(mine))])
        keep = np.concatenate([res.keep_mask(k).astype(np.float64) for k in range(len(mine))])
        svs = np.concatenate([res.singular_values(k) for k in range(len(mine))])
    else:
        lines, keep, svs = np.zeros(0), np.zeros(0), np.zeros(0)
    # the one collective of the path: packed [lines | keep | sv] per rank
    packed = _gather_varlen(np.concatenate([lines, keep, svs]), dist, group, device)
    line_lists, infos = [None] * len(ms), [None] * len(ms)
    for r, idx in enumerate(parts):
        nl = int(sum(ls[i] for i in idx))
        buf = packed[r]
        rl, rk, rs = buf[:4 * nl].reshape(nl, 4), buf[4 * nl:5 * nl] > 0.5, buf[5 * nl:]
        o = so = 0
        for i in idx:
            ll_i = rl[o:o + ls[i]]
            if filter_invalid_features:
                ll_i = ll_i[rk[o:o + ls[i]]]
            line_lists[i] = ll_i.copy()
            infos[i] = KbdmInfo(m=ms[i], l=ls[i], p=p, q=q, singular_values=rs[so:so + ms[i]].copy())
            o += ls[i]
            so += ms[i]
    out_l, out_i = [], []
    for ll_i, info in zip(line_lists, infos):
        if len(ll_i) > 0:                      # reference sampling.py:67-70
            out_l.append(ll_i)
            out_i.append(info)
    return out_l, out_i
