"""CPU: the N>1 path (sharding + the one variable-length gather) with world size 2, over the package's own
standard-library launcher (llckbdm_amd.launch) and over a torch.distributed gloo group.
The per-rank solver is injected: here it is the oracle (test infrastructure), on the GPU box
it is the HIP engine."""
import os
import socket
import sys
import time

import numpy as np
import pytest

from llckbdm_amd.distributed import shard_items

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_items_balanced_and_complete():
    ms = np.arange(100, 401, 2)
    parts = shard_items(ms.astype(float) ** 3, 8)
    allidx = np.sort(np.concatenate(parts))
    assert np.array_equal(allidx, np.arange(len(ms)))
    loads = np.array([(ms[p].astype(float) ** 3).sum() for p in parts])
    assert loads.max() / loads.mean() < 1.02
    assert shard_items([5.0], 4)[0].tolist() == [0] and all(len(x) == 0 for x in shard_items([5.0], 4)[1:])


@pytest.mark.parametrize("cfg", ["C4", "C5"])
def test_shard_tables_of_the_full_multi_gpu_configs_at_world_8(cfg):
    """BASELINE.json configs[3] and [4] at their FULL item lists over 8 ranks (no solve): every item dealt exactly once, the
    sum of m^3 balanced to 2 %, and the packed-block table every rank derives on its own - sizes, 16-byte alignment and
    offsets of the gathered buffer - consistent with the per-rank line / singular-value counts."""
    from llckbdm_amd import datasets
    from llckbdm_amd.distributed import packed_bytes
    if cfg == "C4":
        ms = np.arange(200, 1201, dtype=np.int64)
        sidx = np.zeros(len(ms), dtype=np.int64)
    else:
        mm = np.arange(128, 384, dtype=np.int64)
        ms = np.tile(mm, 64)
        sidx = np.repeat(np.arange(64), len(mm))
    assert len(ms) == (1001 if cfg == "C4" else 16384)
    world = 8
    parts = shard_items(ms.astype(np.float64) ** 3, world)
    again = shard_items(ms.astype(np.float64) ** 3, world)
    assert all(np.array_equal(a, b) for a, b in zip(parts, again))                      # deterministic: no size exchange needed
    allidx = np.sort(np.concatenate(parts))
    assert np.array_equal(allidx, np.arange(len(ms)))                                   # every item exactly once
    loads = np.array([(ms[p].astype(np.float64) ** 3).sum() for p in parts])
    assert loads.max() / loads.mean() <= 1.02, loads.max() / loads.mean()
    sizes = np.array([packed_bytes(ms[p].sum(), ms[p].sum(), len(p)) for p in parts], dtype=np.int64)
    assert (sizes % 16 == 0).all() and (sizes > 0).all()
    offs = np.concatenate([[0], np.cumsum(sizes)])
    for r, p in enumerate(parts):
        L = int(ms[p].sum())
        raw = 32 * L + 8 * L + 4 * len(p) + L
        assert raw + 16 <= sizes[r] < raw + 32 and offs[r] % 16 == 0                     # + the 16-byte trailer
        assert np.all(np.diff(p) > 0)                                                    # a rank keeps the caller's order
        assert len(np.unique(sidx[p])) >= 1
    assert offs[-1] == sizes.sum()
    total = 41 * int(ms.sum()) + 4 * len(ms) + 16 * world
    assert total <= offs[-1] < total + 16 * world


class _OracleResult:
    def __init__(self, lls, svs, status):
        self.lls, self.svs, self.status = lls, svs, status

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
    # KBDM_TEST_STATUS: status word that the "solver" reports for every member with m == 33 (exercises the contract)
    st = int(os.environ.get("KBDM_TEST_STATUS", "0"))
    return _OracleResult(lls, svs, np.array([st if m == 33 else 0 for m in ms], dtype=np.int32))


M_RANGE = [40, 64, 33, 100, 80]


def _worker(rank, world, port, q, transport):
    """One rank: the sharded samplers over `transport` ("tcp": launch.Rendezvous, standard library only; "gloo":
    the same HostComm interface over a torch.distributed gloo group)."""
    from oracle import kbdm_oracle as O
    from llckbdm_amd.distributed import HostComm, sample_kbdm_sharded, sample_kbdm_signals_sharded
    if transport == "tcp":
        from llckbdm_amd.launch import Rendezvous
        rdzv = Rendezvous(rank, world, port)
    else:
        from tests.gloo_rendezvous import GlooRendezvous
        rdzv = GlooRendezvous(rank, world, port)
    try:
        sig = O.make_noisy(O.brain_sim_signal(512), 1e-3, 0)
        comm = HostComm(_oracle_solve, rdzv)
        lls, infos = sample_kbdm_sharded(sig, 5e-4, M_RANGE, p=1, l=None, q=0, comm=comm)
        rl, ri = sample_kbdm_sharded(sig, 5e-4, M_RANGE, p=1, l=30, q=0, comm=comm, root=-1)
        # gather to one root only: the other rank gets nothing
        root_only = sample_kbdm_sharded(sig, 5e-4, M_RANGE, p=1, l=None, q=0, comm=comm, root=1)
        assert (root_only[0] is None) == (rank != 1)
        # a grid of signals (two voxels, their own member lists): item order and the item index survive the sharding
        sigs = np.stack([sig, O.make_noisy(O.brain_sim_signal(512), 1e-3, 1)])
        sidx = [0, 1, 1, 0, 1, 0]
        mlist = [40, 64, 33, 100, 80, 51]
        gl, gi, gx = sample_kbdm_signals_sharded(sigs, 5e-4, sidx, mlist, p=1, l=None, q=0, comm=comm)
        # the status word travels with the blocks: a hard flag raises on every receiving rank, a weak one warns
        import warnings
        os.environ["KBDM_TEST_STATUS"] = "2"
        try:
            sample_kbdm_sharded(sig, 5e-4, M_RANGE, p=1, l=None, q=0, comm=comm)
            raised = False
        except np.linalg.LinAlgError as e:
            raised = "member(s) [2]" in str(e)
        os.environ["KBDM_TEST_STATUS"] = "4"
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            sample_kbdm_sharded(sig, 5e-4, M_RANGE, p=1, l=None, q=0, comm=comm)
        warned = any("weak" in str(x.message) for x in w)
        os.environ["KBDM_TEST_STATUS"] = "0"
        t = rdzv.max(float(rank))
        q.put((rank, [x.tolist() for x in lls], [i.m for i in infos], [x.tolist() for x in rl],
               [i.singular_values.tolist() for i in ri], [x.tolist() for x in gl], [i.m for i in gi], gx, raised, warned, t))
    finally:
        rdzv.close()


@pytest.mark.parametrize("transport", ["tcp", "gloo"])
def test_sharded_sampler_world2(transport):
    import multiprocessing as mp
    from oracle import kbdm_oracle as O
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, transport)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sig = O.make_noisy(O.brain_sim_signal(512), 1e-3, 0)
    ref_l, ref_i = O.sample_kbdm(sig, 5e-4, M_RANGE, p=1, l=None, q=0, normalizer="gemm")
    ref_l30, ref_i30 = O.sample_kbdm(sig, 5e-4, M_RANGE, p=1, l=30, q=0, normalizer="gemm")
    sigs = np.stack([sig, O.make_noisy(O.brain_sim_signal(512), 1e-3, 1)])
    grid_ref = [O.filter_samples(O.kbdm(sigs[s], 5e-4, m=m, normalizer="gemm")[0])
                for s, m in zip([0, 1, 1, 0, 1, 0], [40, 64, 33, 100, 80, 51])]
    for rank, lls, ms, l30, sv30, gl, gm, gx, raised, warned, tmax in outs:   # every rank: the complete, ordered result, bit for bit
        assert ms == [i.m for i in ref_i]
        assert len(lls) == len(ref_l)
        for a, b in zip(lls, ref_l):
            assert np.array_equal(np.array(a), b)
        assert len(l30) == len(ref_l30)
        for a, b in zip(l30, ref_l30):
            assert np.array_equal(np.array(a).reshape(-1, 4), b)
        for a, info in zip(sv30, ref_i30):
            assert np.array_equal(np.array(a), info.singular_values)
        assert gx == list(range(6)) and gm == [40, 64, 33, 100, 80, 51]
        for a, b in zip(gl, grid_ref):
            assert np.array_equal(np.array(a), b)
        assert raised and warned and tmax == 1.0


def test_rendezvous_primitives_world3():
    """launch.Rendezvous alone: allgather / bcast / max / barrier over three ranks (threads: it is plain sockets)."""
    import threading
    from llckbdm_amd.launch import Rendezvous
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = {}

    def run(r):
        rz = Rendezvous(r, 3, port)
        try:
            out[r] = (rz.allgather(bytes([r]) * (r + 1)), rz.bcast(b"id-from-1" if r == 1 else b"", src=1), rz.max(r * 1.5),
                      rz.exchange_id(b"x" * 128 if r == 0 else None))
            rz.barrier()
        finally:
            rz.close()
    th = [threading.Thread(target=run, args=(r,)) for r in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join(60)
    for r in range(3):
        assert out[r] == ([b"\x00", b"\x01\x01", b"\x02\x02\x02"], b"id-from-1", 3.0, b"x" * 128)


def test_rendezvous_drops_a_stranger_and_frames_are_plain_bytes(monkeypatch):
    """A local process that connects without the launch token (or with a bad / duplicate rank) is dropped, the real
    ranks still meet; payloads travel as length-prefixed byte strings (nothing is unpickled)."""
    import struct
    import threading
    from llckbdm_amd import launch
    monkeypatch.setenv("KBDM_RDZV_TOKEN", "s3cret")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = {}

    def run(r):
        rz = launch.Rendezvous(r, 2, port, timeout=30)
        try:
            out[r] = rz.allgather(b"payload-%d" % r)
        finally:
            rz.close()
    t0 = threading.Thread(target=run, args=(0,))
    t0.start()
    for hello in (struct.pack("<I", 1) + b"wrong", struct.pack("<I", 7) + b"s3cret", b"\x01"):
        for _ in range(200):
            try:
                c = socket.create_connection(("127.0.0.1", port), timeout=5)
                break
            except OSError:
                time.sleep(0.02)
        launch._send(c, hello)
        c.close()
    t1 = threading.Thread(target=run, args=(1,))
    t1.start()
    t0.join(60)
    t1.join(60)
    assert out == {0: [b"payload-0", b"payload-1"], 1: [b"payload-0", b"payload-1"]}
    assert launch._unpack_parts(launch._pack_parts([b"", b"ab", b"\x00" * 5]), 3) == [b"", b"ab", b"\x00" * 5]
    with pytest.raises(ConnectionError):
        launch._unpack_parts(launch._pack_parts([b"a"]), 2)
    assert "import pickle" not in open(launch.__file__).read() and not hasattr(launch, "pickle")


def test_spawn_tears_the_other_ranks_down_when_one_fails():
    """a rank that dies while its peers wait for it (here: in a rendezvous receive) must not hang the launcher:
    spawn() polls every child, terminates the others on the first non-zero exit and returns that code."""
    from llckbdm_amd.launch import spawn
    prog = ("import os, sys, time\n"
            "from llckbdm_amd.launch import Rendezvous\n"
            "r = int(os.environ['RANK'])\n"
            "rz = Rendezvous(timeout=120)\n"
            "rz.barrier()\n"
            "if r == 1:\n"
            "    sys.exit(7)\n"
            "rz.barrier()\n"              # ranks 0 and 2 now wait for rank 1, which is gone
            "time.sleep(120)\n")
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    t0 = time.time()
    rc = spawn([sys.executable, "-c", prog], 3, env=env)
    assert rc != 0             # rank 1's 7, or the ConnectionError of a peer that noticed first
    assert time.time() - t0 < 60
    assert spawn([sys.executable, "-c", "import os; assert os.environ['KBDM_RDZV_TOKEN']"], 2, env=env) == 0


def test_packed_block_layout_roundtrip():
    """Host packing == the layout include/kbdm_hip.h documents for kbdm_plan_gather; sizes match kbdm_packed_bytes."""
    from llckbdm_amd import _lib
    from llckbdm_amd.distributed import pack_block, packed_bytes, unpack_block
    rng = np.random.default_rng(0)
    for nl, nsv, nb in ((0, 0, 0), (7, 5, 2), (1000, 333, 17)):
        lines, sv = rng.standard_normal((nl, 4)), rng.standard_normal(nsv)
        status, keep = rng.integers(0, 8, nb).astype(np.int32), rng.integers(0, 2, nl).astype(np.uint8)
        blk = pack_block(lines, sv, status, keep)
        assert blk.size == packed_bytes(nl, nsv, nb) and blk.size % 16 == 0
        assert blk.size == _lib.load().kbdm_packed_bytes(nl, nsv, nb)
        a, b, c, d = unpack_block(blk, nl, nsv, nb)
        assert np.array_equal(a, lines) and np.array_equal(b, sv) and np.array_equal(c, status)
        assert np.array_equal(d, keep.astype(bool))
    from llckbdm_amd.distributed import block_trailer, TRAILER_MAGIC
    assert block_trailer(pack_block(lines, sv, status, keep, rank=5, seq=2 ** 40 + 3)) == (TRAILER_MAGIC, 5, 2 ** 40 + 3)


@pytest.mark.gpu
def test_rccl_gather_one_rank_rehearsal_matches_download():
    """The C-ABI gather (pack on the device + the grouped transfer, here with one rank) returns exactly what
    kbdm_plan_download returns, and sample_kbdm_sharded over RcclComm equals sample_kbdm bit for bit."""
    from llckbdm_amd import datasets
    from llckbdm_amd.distributed import RcclComm, packed_bytes, sample_kbdm_sharded, unpack_block
    from llckbdm_amd.engine import Engine
    from llckbdm_amd.sampling import sample_kbdm
    eng = Engine(0)
    sig = datasets.add_noise(datasets.brain_sim_signal(1024), 1e-3, 3)
    ms = np.array([64, 100, 37, 150], dtype=np.int32)
    plan = eng.plan(1, 1024, np.zeros(4, np.int32), ms, ms, p=1, q=0.0, dwell=5e-4)
    plan.upload(sig.reshape(1, -1))
    plan.execute()
    ref = plan.download()
    comm = RcclComm(eng, 1, 0, lambda uid: uid, force=True)      # a real one-rank RCCL communicator
    sizes = np.array([packed_bytes(plan.total_lines, plan.total_sv, plan.B)], dtype=np.int64)
    buf = comm.gather_plan(plan, sizes)
    ll, sv, st, kp = unpack_block(buf, plan.total_lines, plan.total_sv, plan.B)
    assert np.array_equal(ll, ref.lines) and np.array_equal(sv, ref.sv)
    assert np.array_equal(st, ref.status) and np.array_equal(kp, ref.keep.astype(bool))
    a_l, a_i = sample_kbdm_sharded(sig, 5e-4, ms.tolist(), p=1, l=None, q=0, comm=comm)
    b_l, b_i = sample_kbdm(sig, 5e-4, ms.tolist(), p=1, l=None, q=0, engine=eng)
    assert len(a_l) == len(b_l)
    for x, y in zip(a_l, b_l):
        assert np.array_equal(x, y)
    for x, y in zip(a_i, b_i):
        assert np.array_equal(x.singular_values, y.singular_values)
