#!/usr/bin/env python3
"""KBDM ensemble throughput on MI355X: `python bench.py --gpus N --steps K --warmup W`.

Metric (BASELINE.json): KBDM solves/sec over an m-range ensemble, N=2048 complex signal.
Workload at every rank = BASELINE.json configs[1] ("C2"): 16-peak brain-sim signal + seeded
sigma=1e-3 noise, members m = 100..400 step 2 (151 solves), l = m, p = 1, q = 0.
One "step" = one pass of the whole pipeline (Hankel -> SVD -> reduced eig -> line lists) over
that batch, signals already resident in HBM, line lists left in HBM.  With N > 1 every rank
solves its own ensemble (another noise seed: weak scaling, members are independent) and the
packed line lists are gathered to every rank with ONE RCCL all_gather inside the timed region.

The JSON line also carries
  roofline     : the dominant kernel's algorithmic FP64 flops / its HIP-event duration vs the
                 FP64 matrix peak (flop model: SURVEY.md 8d, stated in DESIGN.md)
  cpu_baseline : the numpy/scipy oracle (the reference's own LAPACK calls) timed on the host
                 cores of this box on a bounded sample of the same workload (rank 0, N=1 only).
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-flight", type=int, default=3, choices=[1, 2, 3, 4],
                    help="ensembles (steps) in flight at once (each has its own plan and three streams; every stream "
                         "needs a hardware queue of its own: GPU_MAX_HW_QUEUES, 16 asked for here)")
    ap.add_argument("--workload", default="C2", choices=["C2", "C3small"])
    ap.add_argument("--no-stagger", dest="stagger", action="store_false",
                    help="submit the second ensemble at once instead of when the first reaches its QR iteration")
    args = ap.parse_args()

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
    if world > 1 or args.gpus > 1 or os.environ.get("KBDM_BENCH_FORCE_DIST") == "1":
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
    else:
        try:
            import torch
        except Exception:
            torch = None

    from llckbdm_amd import datasets
    from llckbdm_amd.engine import Engine

    def init_dist():
        # after the solver's streams exist: the runtime hands out hardware queues in order of stream creation,
        # and the communicator's streams (idle most of the time) should be the ones that share
        if dist is not None:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # Ensembles in flight: every step solves one whole ensemble (its own plan, workspace and streams); a step
    # is submitted without waiting for the previous one, and a plan is waited for only when its workspace is
    # needed again.  One ensemble alone is latency bound (a chain of one-CU-per-member kernels, most of the
    # 256 CUs idle); two or three in flight overlap those chains.  `step_latency_ms` is the single-ensemble latency.
    nfl = max(1, args.in_flight)
    dwell = datasets.DWELL
    engines, plans = [], []
    for k in range(nfl):
        if args.workload == "C2":
            sigs, sig_idx, ms = datasets.config2(seed=rank + 1000 * k)
            wname = "C2: N=2048, 16 peaks + sigma=1e-3 noise, m=100..400:2 (151 members), l=m, p=1, q=0"
        else:
            sigs, sig_idx, ms = datasets.config3(count=64, m=512, seed0=1000 * rank + 100000 * k)
            wname = "C3small: N=2048, 64 pseudo-noise draws (sigma=1e-6), m=512"
        e = Engine(local_rank)
        pk = e.plan(sigs.shape[0], sigs.shape[1], sig_idx, ms, ms, p=1, q=0.0, dwell=dwell)
        pk.upload(sigs)
        engines.append(e)
        plans.append(pk)
    plan = plans[0]
    init_dist()
    sigs0 = datasets.config2(seed=rank)[0] if args.workload == "C2" else None
    units = len(ms)

    gather_buf = local_buf = None
    if dist is not None:
        local_buf = torch.empty((plan.total_lines, 4), dtype=torch.float64, device="cuda")
        gather_buf = torch.empty((world * plan.total_lines, 4), dtype=torch.float64, device="cuda")

    def sync_all():
        for pk in plans:
            pk.sync()
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()

    stage_acc = {}
    busy = [False] * nfl
    timed = [False]

    def finish(k):
        """Complete the step that plan k is running: wait, (multi-process) gather its line lists, stage timers."""
        if not busy[k]:
            return
        pk = plans[k]
        if dist is not None:
            pk.copy_lines_to_device(local_buf.data_ptr(), local_buf.numel() * 8)   # syncs the plan's stream
            dist.all_gather_into_tensor(gather_buf, local_buf)
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
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ok = min(int((pk.download().status == 0).sum()) for pk in plans)

    # the same steps one at a time (outside the timed region; reported next to the headline for comparison)
    serial = None
    if nfl > 1 and dist is None:
        timed[0] = False
        ns_ser = min(args.steps, 5)
        ts = time.perf_counter()
        for s in range(ns_ser):
            plans[0].execute(sync=True)
        ts = time.perf_counter() - ts
        serial = {"value": units * ns_ser / ts, "ms_per_step": 1e3 * ts / ns_ser, "steps": ns_ser, "ensembles_in_flight": 1}

    if rank == 0:
        value = world * units * args.steps / elapsed
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
        # HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/), if any
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r1_last_pmc_traffic.json")))["kernels"]
            # lane 0 runs the QR iteration as k_hqr_team; k_gen is templated on the register chunk count
            for key in (dom + "_team", dom, dom.split("(")[0], dom.split("(")[0] + "<8>"):
                if key in pmc:
                    traffic = pmc[key].get("hbm_bytes_per_launch")
                    break
        except Exception:
            traffic = None
        roofline = {"kernel": dom, "bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic,
                    "avg_ms": stage_ms[dom], "algorithmic_flops_per_launch": fl[dom],
                    "launch_members": n0}
        total_fl = sum(fl_all.values())
        out = {
            "metric": "KBDM solves/sec over m-range ensemble, N=2048 complex signal",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64 (complex128)", "data": "synthetic",
            "config": {"workload": wname, "members_per_gpu": units, "parallelism": f"ensemble-sharded x{world}",
                       "ensembles_in_flight": nfl,
                       "collective": "one all_gather of packed line lists (RCCL)" if world > 1 else "none"},
            "roofline": roofline,
            "pipeline_tflops": total_fl * args.steps * world / elapsed / 1e12,
            "stage_ms": stage_ms,
            "step_latency_ms": None if latency is None else 1e3 * latency,
            "one_ensemble_at_a_time": serial,
            "members_ok": ok,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(sigs0[0], ms, dwell) if args.workload == "C2" else None
            except Exception as e:   # the baseline is informational; never lose the GPU number over it
                out["cpu_baseline"] = {"error": repr(e)}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
