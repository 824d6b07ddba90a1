#!/usr/bin/env python3
"""KBDM ensemble throughput on MI355X: `python bench.py --gpus N --steps K --warmup W`.

Metric (BASELINE.json): KBDM solves/sec over an m-range ensemble, N=2048 complex signal.
Workload at every rank = BASELINE.json configs[1] ("C2"): 16-peak brain-sim signal + seeded
sigma=1e-3 noise, members m = 100..400 step 2 (151 solves), l = m, p = 1, q = 0.
One "step" = one pass of the whole pipeline (Hankel -> SVD -> reduced eig -> line lists) over
that batch, signals already resident in HBM, line lists left in HBM.  With N > 1 every rank
solves its own ensemble (another noise seed: weak scaling, members are independent) and the
packed results are gathered to every rank with ONE grouped RCCL transfer (kbdm_plan_gather of the
C ABI: device buffers to device buffers over xGMI) inside the timed region.  `--sharded` instead
deals ONE ensemble (default C4: 1001 members, N=4096) over the ranks (strong scaling, the
partition of llckbdm_amd.distributed.shard_items) with the same gather.  torch.distributed is the
process launcher and the control plane only (gloo: barrier, max over ranks, id hand-off).
`python bench.py --gpus N` without a launcher starts its own ranks (torch.distributed.run).

The JSON line also carries
  roofline     : the dominant kernel's algorithmic FP64 flops / its HIP-event duration vs the
                 FP64 matrix peak (flop model: SURVEY.md 8d, stated in DESIGN.md)
  cpu_baseline : the numpy/scipy oracle (the reference's own LAPACK calls) timed on the host
                 cores of this box on a bounded sample of the same workload (rank 0, N=1 only):
                 one process per core (best effort) and, as `serial`, the shape the reference ships
                 (sampling.py:52-62: one loop, default BLAS threads).
  host_inclusive : the same ensemble from host signals to host line lists (plan reused: H2D +
                 execute + D2H), SURVEY.md 8d's wording of the metric.
  north_star_workload : the m = 100..500 ensemble BASELINE.json's target is quoted on.
"""
import argparse
import json
import os
import sys
import time

# The pipeline runs on three HIP streams; torch and RCCL add their own.  With the runtime's default of four
# hardware queues those streams would share queues (measured: 194 instead of 175 ms per step in the
# multi-process path), so ask for eight before anything initialises HIP.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6       # MI355X FP64 matrix (= vector) peak, datasheet; see DESIGN.md
HBM_PEAK_GBS = 8000.0


def stage_flops(m, l):
    """Algorithmic real flops per member and pipeline stage (SURVEY.md 8d split of 216 m^3)."""
    m, l = float(m), float(l)
    return {
        "k_hankel": 0.0,
        "k_svd_fac": (32.0 / 3.0) * m ** 3,          # Householder bidiagonalisation
        "k_gen(Q,P)": (32.0 / 3.0) * m ** 3,         # explicit Q and P
        "k_bdsqr_gen": 0.0,                          # O(m^2) scalar recurrence
        "k_bdsqr_apply": (84.0 - 64.0 / 3.0) * m ** 3,   # remainder of the 84 m^3 SVD budget
        "k_bdsqr_sort": 0.0,
        "k_gemm<1>": 8.0 * m * m * l,
        "k_gemm<2>": 8.0 * l * l * m,
        "k_hess": (40.0 / 3.0) * l ** 3,             # Hessenberg reduction
        "k_gen(Qh)": (16.0 / 3.0) * l ** 3,          # explicit Qh
        "k_hqr": (100.0 - 56.0 / 3.0 - 16.0) * l ** 3,
        "k_invit": 8.0 * l ** 3,
        "k_gemm<3>": 8.0 * l ** 3,
        "k_gemm<4>": 8.0 * m * l * l,
        "k_gemm<5>": 8.0 * m * m * l,
        "k_epilogue": 0.0,
    }


def _cpu_worker(args):
    sig, m, dwell = args
    os.environ["OPENBLAS_NUM_THREADS"] = "1"
    from oracle import kbdm_oracle as O
    ll, _ = O.kbdm(sig, dwell, m=int(m), normalizer="gemm")
    return len(O.filter_samples(ll))


def cpu_baseline(sig, ms, dwell, min_seconds=12.0, max_passes=8):
    """Oracle (numpy/scipy: the reference's own LAPACK calls) over the WHOLE C2 member list,
    one process per host core with 1 BLAS thread each, repeated until >= min_seconds of wall."""
    import multiprocessing as mp
    from threadpoolctl import threadpool_limits
    members = [int(m) for m in ms][::-1]            # longest first: better balance
    # the GPU box gives each GPU a 16-core host share
    cores = max(1, min(len(members), 16, len(os.sched_getaffinity(0)), (os.cpu_count() or 1)))
    with threadpool_limits(1):
        ctx = mp.get_context("fork")
        with ctx.Pool(cores) as pool:
            pool.map(_cpu_worker, [(sig, 32, dwell)] * cores)            # warm imports
            t0 = time.perf_counter()
            passes = 0
            while passes < max_passes and (time.perf_counter() - t0) < min_seconds:
                pool.map(_cpu_worker, [(sig, m, dwell) for m in members], chunksize=1)
                passes += 1
            dt = time.perf_counter() - t0
    return {"value": passes * len(members) / dt, "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": f"{passes} pass(es) over all {len(members)} members of C2, numpy/scipy oracle "
                      f"(zgesdd + zgeev + gemm normaliser), one process per core x {cores}, "
                      f"1 BLAS thread each, {dt:.1f} s wall"}


def cpu_baseline_serial(sig, ms, dwell, stride=5):
    """SURVEY.md 8d mode (i): the loop as the reference ships it (sampling.py:52-62): one process, members one
    after the other, default BLAS threading - on every `stride`-th member of the workload."""
    from oracle import kbdm_oracle as O
    members = [int(m) for m in ms][::stride]
    O.kbdm(sig, dwell, m=32, normalizer="gemm")
    t0 = time.perf_counter()
    for m in members:
        O.kbdm(sig, dwell, m=m, normalizer="gemm")
    dt = time.perf_counter() - t0
    return {"value": len(members) / dt, "unit": "solves/s", "cores": len(os.sched_getaffinity(0)), "kind": "port",
            "sample": f"every {stride}th member of the workload ({len(members)} solves) in one serial loop with default "
                      f"BLAS threads, as sampling.py:52-62 ships it, {dt:.1f} s wall"}


def _self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks as child processes (never re-exec this one:
    nothing here has touched the GPU yet, but a child keeps the contract simple) and forward rank 0's line."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 50 steps by default: the timed region is synchronised on both sides, so it contains the fill and the drain of the
    # three-deep pipeline (about 45 ms each); at ~60 ms per step that is 3 % of the region (7 % with 20 steps)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-flight", type=int, default=3, choices=[1, 2, 3, 4],
                    help="ensembles (steps) in flight at once (each has its own plan and three streams; every stream "
                         "needs a hardware queue of its own: GPU_MAX_HW_QUEUES, 16 asked for here)")
    ap.add_argument("--workload", default=None, choices=["C2", "NS", "C3small", "C4"],
                    help="C2 (default), NS = north-star ensemble m=100..500, C3small, C4 (default of --sharded)")
    ap.add_argument("--sharded", action="store_true",
                    help="strong scaling: ONE ensemble dealt over the ranks (shard_items) + the RCCL gather per step")
    ap.add_argument("--no-extras", action="store_true", help="skip host_inclusive / north_star_workload")
    ap.add_argument("--no-stagger", dest="stagger", action="store_false",
                    help="submit the second ensemble at once instead of when the first reaches its QR iteration")
    args = ap.parse_args()
    if args.workload is None:
        args.workload = "C4" if args.sharded else "C2"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_launch(args))

    # stdout carries exactly one line (the JSON): anything a library prints there while we run (RCCL's version
    # banner at communicator creation, for one) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    torch = None
    # KBDM_BENCH_FORCE_DIST=1: take the multi-process path (RCCL init, device-side gather) even with one rank -
    # a rehearsal of the N > 1 code on a one-GPU box
    force_dist = os.environ.get("KBDM_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch
        import torch.distributed as dist

    from llckbdm_amd import datasets
    from llckbdm_amd.engine import Engine

    comm_box = [None]

    def init_dist():
        # after the solver's streams exist: the runtime hands out hardware queues in order of stream creation,
        # and the communicator's streams (idle most of the time) should be the ones that share.  torch.distributed
        # (gloo, CPU) is the control plane; the data path is the library's own RCCL communicator.
        if dist is None:
            return
        if world > 1 or "MASTER_ADDR" in os.environ:
            dist.init_process_group("gloo")
        else:                                   # one-rank rehearsal without a launcher
            dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1)
        from llckbdm_amd.distributed import RcclComm

        def exchange(uid):
            box = [uid]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        comm_box[0] = RcclComm(engines[0], world, rank, exchange, force=True)

    # Ensembles in flight: every step solves one whole ensemble (its own plan, workspace and streams); a step
    # is submitted without waiting for the previous one, and a plan is waited for only when its workspace is
    # needed again.  One ensemble alone is latency bound (a chain of one-CU-per-member kernels, most of the
    # 256 CUs idle); two or three in flight overlap those chains.  `step_latency_ms` is the single-ensemble latency.
    nfl = max(1, args.in_flight)
    dwell = datasets.DWELL
    from llckbdm_amd.distributed import packed_bytes, shard_items

    def workload(seed):
        if args.workload == "C2":
            return datasets.config2(seed=seed) + ("C2: N=2048, 16 peaks + sigma=1e-3 noise, m=100..400:2 (151 members), l=m, p=1, q=0",)
        if args.workload == "NS":
            return datasets.north_star(seed=seed) + ("NS: N=2048, 16 peaks + sigma=1e-3 noise, m=100..500:2 (201 members), l=m, p=1, q=0",)
        if args.workload == "C4":
            return datasets.config4() + ("C4: N=4096, 32 peaks + sigma=1e-3 noise, m=200..1200 (1001 members), l=m, p=1, q=0",)
        return datasets.config3(count=64, m=512, seed0=seed) + ("C3small: N=2048, 64 pseudo-noise draws (sigma=1e-6), m=512",)

    # memory of one plan of this rank's share (five m x m work buffers + the rotation log, ~176 m^2 bytes per member): cap the
    # ensembles in flight so that all plans fit the GPU (the full C4 on one or two ranks: one in flight)
    if args.sharded:
        _, _, ms_probe, _ = workload(0)
        share = ms_probe[shard_items(ms_probe.astype(np.float64) ** 3, world)[rank]].astype(np.float64)
        est = float(np.sum(176.0 * share ** 2)) + 2e9
        nfl = max(1, min(nfl, int(200e9 // est)))
    engines, plans, gather_sizes = [], [], []
    for k in range(nfl):
        if args.sharded:
            # ONE ensemble per step, its members dealt over the ranks (every rank derives the same table)
            sigs, sig_idx, ms_all, wname = workload(1000 * k)
            parts = shard_items(ms_all.astype(np.float64) ** 3, world)
            mine = parts[rank]
            ms, sidx = ms_all[mine], sig_idx[mine]
            gather_sizes.append(np.array([packed_bytes(ms_all[ix].sum(), ms_all[ix].sum(), len(ix)) for ix in parts],
                                         dtype=np.int64))
            units = len(ms_all)
        else:
            sigs, sidx, ms, wname = workload(rank + 1000 * k)
            gather_sizes.append(np.array([packed_bytes(ms.sum(), ms.sum(), len(ms))] * world, dtype=np.int64))
            units = len(ms)
        e = Engine(local_rank)
        pk = e.plan(sigs.shape[0], sigs.shape[1], sidx, ms, ms, p=1, q=0.0, dwell=dwell)
        pk.upload(sigs)
        engines.append(e)
        plans.append(pk)
    plan = plans[0]
    init_dist()
    comms = []
    if dist is not None:
        from llckbdm_amd.distributed import RcclComm

        def exchange(uid):
            box = [uid]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        comms = [comm_box[0]] + [RcclComm(engines[k], world, rank, exchange, force=True) for k in range(1, nfl)]
    sigs0 = workload(rank)[0] if args.workload in ("C2", "NS") else None

    def sync_all():
        for pk in plans:
            pk.sync()

    stage_acc = {}
    busy = [False] * nfl
    timed = [False]

    def finish(k):
        """Complete the step that plan k is running: wait, (multi-process) gather its line lists, stage timers."""
        if not busy[k]:
            return
        pk = plans[k]
        if dist is not None:
            # the one collective of the path: every rank's packed block, device to device (waits for the plan)
            comms[k].engine.lib.kbdm_plan_gather(pk.handle, world, rank, gather_sizes[k].ctypes.data, -1, None)
        if timed[0]:
            for name, v in pk.stage_ms().items():      # HIP events of the critical lane (waits for the plan)
                stage_acc[name] = stage_acc.get(name, 0.0) + v
        else:
            pk.sync()
        busy[k] = False

    def step(s):
        k = s % nfl
        finish(k)
        if timed[0] and 1 <= s < nfl and args.stagger:
            # the pipelines have the same cycle: started together they stay in phase (panels against panels,
            # QR iteration against QR iteration); start each one a fraction of the cycle after the one before it
            # (two in flight: when the first reaches its QR iteration)
            plans[s - 1].wait_stage({2: "k_hess", 3: "k_bdsqr_sort"}.get(nfl, "k_svd_fac"))
        plans[k].execute(sync=False)
        busy[k] = True

    # warm-up: one ensemble at a time, which also gives the single-ensemble latency
    latency = None
    for s in range(args.warmup):
        sync_all()
        tw = time.perf_counter()
        step(s)
        finish(s % nfl)
        sync_all()
        tw = time.perf_counter() - tw
        latency = tw if latency is None else min(latency, tw)
    for k in range(nfl):               # every plan has run once before the timed region
        if args.warmup <= k:
            step(k)
            finish(k)
    sync_all()
    if dist is not None:
        dist.barrier()
    sync_all()
    timed[0] = True
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(s)
    for s in range(args.steps, args.steps + nfl):     # drain in submission order
        finish(s % nfl)
    sync_all()
    if dist is not None:
        dist.barrier()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ok = min(int((pk.download().status == 0).sum()) for pk in plans)
    if dist is not None and world == 1:
        # one-rank rehearsal: the gathered block must be exactly the plan's results
        from llckbdm_amd.distributed import unpack_block
        buf = comms[0].gather_plan(plans[0], gather_sizes[0])
        ref = plans[0].download()
        ll, sv, st, kp = unpack_block(buf, plans[0].total_lines, plans[0].total_sv, plans[0].B)
        assert np.array_equal(ll, ref.lines) and np.array_equal(sv, ref.sv) and np.array_equal(kp, ref.keep.astype(bool))

    # the same steps one at a time (outside the timed region; reported next to the headline for comparison)
    serial = host_incl = ns_line = None
    if dist is None:
        timed[0] = False
        ns_ser = min(args.steps, 5)
        if nfl > 1:
            ts = time.perf_counter()
            for s in range(ns_ser):
                plans[0].execute(sync=True)
            ts = time.perf_counter() - ts
            serial = {"value": units * ns_ser / ts, "ms_per_step": 1e3 * ts / ns_ser, "steps": ns_ser, "ensembles_in_flight": 1}
        if not args.no_extras:
            # SURVEY.md 8d's wording of the metric: host signals resident -> host line lists resident, with the plan
            # (device workspace) reused as Engine.solve does for a repeated geometry
            sig_host = np.ascontiguousarray(workload(rank)[0])
            th = time.perf_counter()
            for s in range(ns_ser):
                plans[0].upload(sig_host)
                plans[0].execute(sync=False)
                plans[0].download()
            th = time.perf_counter() - th
            host_incl = {"value": units * ns_ser / th, "unit": "solves/s", "ms_per_step": 1e3 * th / ns_ser, "steps": ns_ser,
                         "ensembles_in_flight": 1, "includes": "H2D of the signal + execute + D2H of lines, sv, mu, keep, status"}
            if args.workload == "C2" and not args.sharded:
                # the ensemble the north-star target is quoted on (m = 100..500), same number of ensembles in flight
                for pk in plans[1:]:
                    pk.close()
                ns_plans = []
                for k in range(nfl):
                    sg, si, mm = datasets.north_star(seed=rank + 1000 * k)
                    pk = engines[k].plan(1, sg.shape[1], si, mm, mm, p=1, q=0.0, dwell=dwell)
                    pk.upload(sg)
                    pk.execute(sync=True)
                    ns_plans.append(pk)
                nst = max(nfl, min(args.steps, 8 * nfl))     # (fill and drain of the pipeline are inside this region too)
                tn = time.perf_counter()
                for s in range(nst):
                    if s >= nfl:
                        ns_plans[s % nfl].sync()
                    ns_plans[s % nfl].execute(sync=False)
                for pk in ns_plans:
                    pk.sync()
                tn = time.perf_counter() - tn
                t1 = time.perf_counter()
                ns_plans[0].execute(sync=True)
                t1 = time.perf_counter() - t1
                ns_ok = int((ns_plans[0].download().status == 0).sum())
                ns_line = {"workload": "N=2048, 16 peaks + sigma=1e-3 noise, m=100..500:2 (201 members)",
                           "value": len(mm) * nst / tn, "unit": "solves/s", "ms_per_step": 1e3 * tn / nst, "steps": nst,
                           "ensembles_in_flight": nfl, "one_ensemble_at_a_time": len(mm) / t1, "members_ok": ns_ok}

    if rank == 0:
        value = (1 if args.sharded else world) * units * args.steps / elapsed
        stage_ms = {k: v / args.steps for k, v in stage_acc.items()}
        # stage timers are those of lane 0 (the largest members, the critical path): price its launches
        # with the flops of exactly those members; `pipeline_tflops` below uses all members
        n0 = plan.lane0_members()
        lane0 = sorted((int(m) for m in ms), reverse=True)[:n0]
        fl, fl_all = {}, {}
        for m in lane0:
            for k, v in stage_flops(m, m).items():
                fl[k] = fl.get(k, 0.0) + v
        for m in ms:
            for k, v in stage_flops(m, m).items():
                fl_all[k] = fl_all.get(k, 0.0) + v
        dom = max(stage_ms, key=lambda k: stage_ms[k])
        achieved = fl[dom] / (stage_ms[dom] * 1e-3) / 1e12 if stage_ms[dom] > 0 else 0.0
        # the kernel behind the dominant stage timer (lane 0 runs the QR iteration as the team kernel)
        hqr_v2 = os.environ.get("KBDM_HQR_V", "2") != "1"
        kname = {"k_hqr": "k_hqr2_team" if hqr_v2 else "k_hqr_team", "k_svd_fac": "k_bidiag_panel<0>", "k_hess": "k_hess_panel",
                 "k_gen(Q,P)": "k_gen<8>", "k_bdsqr_apply": "k_bdsqr_stream", "k_invit": "k_invit_reg<8>"}.get(dom, dom)
        # HBM bytes per launch of that kernel: NOT measured by this run - taken from the newest committed PMC passes
        # (profiles/*_pmc_traffic.json, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this same command), if any
        traffic = traffic_src = None
        try:
            import glob
            for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
                pmc = json.load(open(fn))["kernels"]
                if kname in pmc:
                    traffic = pmc[kname].get("hbm_bytes_per_launch")
                    traffic_src = "profiles/" + os.path.basename(fn) + " (committed rocprofv3 PMC passes, not this run)"
                    break
        except Exception:
            traffic = None
        roofline = {"kernel": kname, "stage_timer": dom,
                    "bound": "mfma" if kname in ("k_trail_update", "k_hess_update") else "fp64_vector",
                    "bound_note": "FP64 vector FMA, instruction-issue / latency bound (no MFMA in this kernel); "
                                  "MI355X FP64 vector peak = FP64 matrix peak = 78.6 TFLOP/s",
                    "achieved": achieved, "peak": FP64_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                    "avg_ms": stage_ms[dom], "algorithmic_flops_per_launch": fl[dom],
                    "launch_members": n0}
        total_fl = sum(fl_all.values())
        out = {
            "metric": "KBDM solves/sec over m-range ensemble, N=2048 complex signal",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong" if args.sharded else "weak",
            "vs_baseline": None, "dtype": "f64 (complex128)", "data": "synthetic",
            "config": {"workload": wname, "members_per_gpu": len(ms), "members_per_step": units * (1 if args.sharded else world),
                       "parallelism": (f"one ensemble sharded over {world} rank(s) by LPT on m^3" if args.sharded
                                       else f"one ensemble per rank x{world}"),
                       "ensembles_in_flight": nfl,
                       "collective": ("one grouped RCCL send/recv of the packed results per step (kbdm_plan_gather)"
                                      if dist is not None else "none")},
            "roofline": roofline,
            "pipeline_tflops": total_fl * args.steps * world / elapsed / 1e12,
            "stage_ms": stage_ms,
            "step_latency_ms": None if latency is None else 1e3 * latency,
            "one_ensemble_at_a_time": serial,
            "host_inclusive": host_incl,
            "north_star_workload": ns_line,
            "members_ok": ok,
        }
        if world == 1 and dist is None and not args.no_cpu_baseline:
            try:
                if args.workload in ("C2", "NS"):
                    out["cpu_baseline"] = cpu_baseline(sigs0[0], ms, dwell)
                    out["cpu_baseline"]["serial"] = cpu_baseline_serial(sigs0[0], ms, dwell)
                else:
                    out["cpu_baseline"] = None
            except Exception as e:   # the baseline is informational; never lose the GPU number over it
                out["cpu_baseline"] = {"error": repr(e)}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        for c in comms:
            c.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
