"""CPU: the N>1 path (sharding + the one variable-length gather) on world_size=2 gloo.
The per-rank solver is injected: here it is the oracle (test infrastructure), on the GPU box
it is the HIP engine."""
import os
import socket

import numpy as np
import pytest

from llckbdm_amd.distributed import shard_items


def test_shard_items_balanced_and_complete():
    ms = np.arange(100, 401, 2)
    parts = shard_items(ms.astype(float) ** 3, 8)
    allidx = np.sort(np.concatenate(parts))
    assert np.array_equal(allidx, np.arange(len(ms)))
    loads = np.array([(ms[p].astype(float) ** 3).sum() for p in parts])
    assert loads.max() / loads.mean() < 1.02
    assert shard_items([5.0], 4)[0].tolist() == [0] and all(len(x) == 0 for x in shard_items([5.0], 4)[1:])


class _OracleResult:
    def __init__(self, lls, svs):
        self.lls, self.svs = lls, svs

    def line_list(self, i):
        return self.lls[i]

    def keep_mask(self, i):
        ll = self.lls[i]
        return (ll[:, 0] > 1e-6) & (ll[:, 1] > 0)

    def singular_values(self, i):
        return self.svs[i]


def _oracle_solve(signals, sig_idx, ms, ls, p, q, dwell):
    from oracle import kbdm_oracle as O
    lls, svs = [], []
    for s, m, l in zip(sig_idx, ms, ls):
        ll, info = O.kbdm(signals[s], dwell, m=m, l=l, p=p, q=q, normalizer="gemm")
        lls.append(ll)
        svs.append(info.singular_values)
    return _OracleResult(lls, svs)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import kbdm_oracle as O
        from llckbdm_amd.distributed import sample_kbdm_sharded
        sig = O.make_noisy(O.brain_sim_signal(512), 1e-3, 0)
        m_range = [40, 64, 33, 100, 80]
        lls, infos = sample_kbdm_sharded(sig, 5e-4, m_range, p=1, l=None, q=0, solve=_oracle_solve)
        q.put((rank, [x.tolist() for x in lls], [i.m for i in infos]))
    finally:
        dist.destroy_process_group()


def test_sharded_sampler_world2_gloo():
    import torch.multiprocessing as mp
    from oracle import kbdm_oracle as O
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sig = O.make_noisy(O.brain_sim_signal(512), 1e-3, 0)
    m_range = [40, 64, 33, 100, 80]
    ref_l, ref_i = O.sample_kbdm(sig, 5e-4, m_range, p=1, l=None, q=0, normalizer="gemm")
    for rank, lls, ms in outs:                 # every rank holds the complete, ordered result
        assert ms == [i.m for i in ref_i]
        assert len(lls) == len(ref_l)
        for a, b in zip(lls, ref_l):
            assert np.array_equal(np.array(a), b)
