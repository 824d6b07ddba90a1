#!/usr/bin/env python3
"""KBDM ensemble throughput on MI355X: `python bench.py --gpus N --steps K --warmup W`.

Metric (BASELINE.json): KBDM solves/sec over an m-range ensemble, N=2048 complex signal.
Default workload at every rank = BASELINE.json configs[1] ("C2"): 16-peak brain-sim signal + seeded
sigma=1e-3 noise, members m = 100..400 step 2 (151 solves), l = m, p = 1, q = 0.
One "step" = one whole ensemble through the pipeline (Hankel -> SVD -> reduced eig -> line lists), submitted
through the package's public scheduler `Engine.submit` / `Pending.result` (four ensembles in flight on four
contexts); `value` is measured with the signals already resident in HBM and the results landing in host memory,
`host_to_host` with the upload inside the loop as well (the unit sampling.py:52-70 defines).
With N > 1 ranks (one process per GPU; the launcher's RANK / LOCAL_RANK / WORLD_SIZE are read from the environment):
  default      every rank solves its own ensembles (another noise seed: weak scaling, members are independent) and
               the packed results are gathered to rank 0 with ONE grouped RCCL transfer per step (kbdm_plan_gather
               of the C ABI: device buffers to device buffers over xGMI) inside the timed region;
  --sharded    ONE job dealt over the ranks by LPT on m^3 (llckbdm_amd.distributed.shard_items) with the same gather:
               --workload C5 (default: 64 voxels x 256 members) or C4 (m = 200..1200, N = 4096) - strong scaling.
The control plane (rank 0's communicator id, barrier, max over ranks) is llckbdm_amd.launch.Rendezvous: one TCP
socket per rank, standard library only.  `python bench.py --gpus N` without a launcher starts its own ranks.

The JSON line also carries
  roofline     : the dominant kernel's algorithmic FP64 flops / its HIP-event duration (events recorded by the library
                 on the stream the kernel runs on) vs the FP64 peak (flop model: SURVEY.md 8d, stated in DESIGN.md)
  cpu_baseline : the numpy/scipy oracle (the reference's own LAPACK calls) timed on the host cores of this box on a
                 bounded sample of the same workload (rank 0, N=1 only): one process per core (best effort) and, as
                 `serial`, the shape the reference ships (sampling.py:52-62: one loop, default BLAS threads).
  other_configs: short runs of the other BASELINE.json configurations (C3, C4, C5) and of the north-star ensemble
                 m = 100..500 through the same public API (N=1 only).
"""
import argparse
import json
import os
import subprocess
import sys
import time

# Three contexts x three HIP streams; RCCL adds its own.  With the runtime's default of four hardware queues those
# streams would share queues, so ask for sixteen before anything initialises HIP.
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
        "k_dc_tree": (16.0 / 3.0) * m ** 3,          # divide and conquer: node products over all levels (4/3 * 4 m^3)
        "k_dc_final": 8.0 * m ** 3,                  # L = Q X, R = P Y (complex x real)
        "k_dc_sv": 0.0,
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


def git_head():
    """HEAD of the work tree, or - on a GPU box, whose snapshot has no .git - the head recorded when the library was built."""
    try:
        h = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                           timeout=10).stdout.strip()
        if h:
            return h
    except Exception:
        pass
    try:
        with open(os.path.join(ROOT, "llckbdm_amd", "build_head.txt")) as f:
            return f.read().strip() or None
    except Exception:
        return None


def file_sha(path):
    import hashlib
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except Exception:
        return None


WORKLOADS = {
    "C2": "C2: N=2048, 16 peaks + sigma=1e-3 noise, m=100..400:2 (151 members), l=m, p=1, q=0",
    "NS": "NS: N=2048, 16 peaks + sigma=1e-3 noise, m=100..500:2 (201 members), l=m, p=1, q=0",
    "C3": "C3: N=2048, 1024 pseudo-noise draws (sigma=1e-6), m=512, l=m, p=1, q=0",
    "C3small": "C3small: N=2048, 64 pseudo-noise draws (sigma=1e-6), m=512",
    "C4": "C4: N=4096, 32 peaks + sigma=1e-3 noise, m=200..1200 (1001 members), l=m, p=1, q=0",
    "C5": "C5: 64 voxels (N=2048, 16 peaks with scaled amplitudes + sigma=1e-3 noise) x m=128..383 (16384 members)",
}


def make_workload(name, seed):
    from llckbdm_amd import datasets
    if name == "C2":
        return datasets.config2(seed=seed)
    if name == "NS":
        return datasets.north_star(seed=seed)
    if name == "C3":
        return datasets.config3(count=1024, m=512, seed0=seed)
    if name == "C3small":
        return datasets.config3(count=64, m=512, seed0=seed)
    if name == "C4":
        return datasets.config4()
    if name == "C5":
        return datasets.config5()
    raise ValueError(name)


def plan_bytes_estimate(ms):
    # five m x m work buffers + the rotation log, ~176 m^2 bytes per member
    return float(np.sum(176.0 * np.asarray(ms, dtype=np.float64) ** 2)) + 2e9


def run_api_loop(eng, works, steps, nfl, resident, on_done=None):
    """`steps` ensembles through Engine.submit with `nfl` in flight; returns elapsed seconds.  works[k] = (signals,
    sig_idx, ms) of the ensemble that step s = k (mod nfl) solves.  on_done(pending) runs when a step is retired."""
    from collections import deque
    pend = deque()
    t0 = time.perf_counter()
    for s in range(steps):
        if len(pend) == nfl:
            h = pend.popleft()
            if on_done:
                on_done(h)
            h.result(check=False)
        sg, si, mm = works[s % nfl]
        pend.append(eng.submit(sg, si, mm, mm, p=1, q=0.0, dwell=DWELL, resident=resident))
    while pend:
        h = pend.popleft()
        if on_done:
            on_done(h)
        h.result(check=False)
    return time.perf_counter() - t0


DWELL = 5e-4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 50 steps by default: the timed region is synchronised on both sides, so it contains the fill and the drain of the
    # three-deep pipeline (about 45 ms each); at ~60 ms per step that is 3 % of the region (7 % with 20 steps)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-flight", type=int, default=4, choices=[1, 2, 3, 4, 5, 6],
                    help="ensembles (steps) in flight at once (Engine(in_flight=...): one context = three streams each; "
                         "every stream needs a hardware queue of its own: GPU_MAX_HW_QUEUES, 16 asked for here)")
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="C2 (default), NS = north-star ensemble m=100..500, C3, C3small, C4, C5 (default of --sharded)")
    ap.add_argument("--sharded", action="store_true",
                    help="strong scaling: ONE job dealt over the ranks (shard_items) + the RCCL gather per step")
    ap.add_argument("--no-extras", action="store_true", help="skip host_to_host / other_configs")
    ap.add_argument("--no-stagger", dest="stagger", action="store_false",
                    help="submit the ensembles of a burst at once instead of a fraction of a cycle apart")
    args = ap.parse_args()
    if args.workload is None:
        args.workload = "C5" if args.sharded else "C2"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from llckbdm_amd.launch import spawn
        sys.exit(spawn([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))

    # stdout carries exactly one line (the JSON): anything a library prints there while we run (RCCL's version
    # banner at communicator creation, for one) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # KBDM_BENCH_FORCE_DIST=1: take the multi-process path (RCCL init, device-side gather) even with one rank -
    # a rehearsal of the N > 1 code on a one-GPU box
    multi = world > 1 or os.environ.get("KBDM_BENCH_FORCE_DIST") == "1"

    from llckbdm_amd import _lib
    from llckbdm_amd.distributed import RcclComm, packed_bytes, shard_items, unpack_block
    from llckbdm_amd.engine import Engine
    from llckbdm_amd.launch import Rendezvous

    rdzv = Rendezvous(rank, world) if multi else None
    wname = WORKLOADS[args.workload]
    big = args.workload in ("C3", "C4", "C5")

    # ---- this rank's work: works[k] = the ensemble a step with k = s mod nfl solves
    nfl = max(1, args.in_flight)
    parts = None
    if args.sharded:
        sigs_all, sidx_all, ms_all = make_workload(args.workload, 0)
        parts = shard_items(ms_all.astype(np.float64) ** 3, world)
        mine = parts[rank]
        used = np.unique(sidx_all[mine])
        remap = np.zeros(sigs_all.shape[0], dtype=np.int32)
        remap[used] = np.arange(len(used), dtype=np.int32)
        my = (np.ascontiguousarray(sigs_all[used]), remap[sidx_all[mine]], ms_all[mine])
        nfl = max(1, min(nfl, int(200e9 // plan_bytes_estimate(my[2])))) if not big else 1
        works = [my] * nfl
        units = len(ms_all)
        sizes = np.array([packed_bytes(ms_all[ix].sum(), ms_all[ix].sum(), len(ix)) for ix in parts], dtype=np.int64)
    else:
        if big:
            nfl = 1
        works = [make_workload(args.workload, rank + 1000 * k) for k in range(nfl)]
        units = len(works[0][2])
        sizes = np.array([packed_bytes(works[0][2].sum(), works[0][2].sum(), units)] * world, dtype=np.int64)
    ms = works[0][2]

    eng = Engine(local_rank, in_flight=nfl)
    eng.stagger = args.stagger
    ctxs = eng.ensure_contexts()
    # the communicators come AFTER the solver's streams: the runtime hands out hardware queues in order of stream
    # creation, and the communicator's streams (idle most of the time) should be the ones that share
    comms = {}
    if multi:
        first = None
        for c in ctxs:       # ONE communicator per process: the other contexts borrow it (gathers are issued in step order)
            comms[c.value] = RcclComm(eng, world, rank, rdzv.exchange_id, force=True, ctx=c, share=first)
            first = first or comms[c.value]

    stage_acc, nstage = {}, [0]
    gather_t = [0.0, 0.0, 0]             # host seconds waiting for the plan / inside the gather call / calls
    timed = [False]
    gathered_ok = [True]

    def on_done(h):
        """Retire a step: (multi-process) the one collective of the path, then the stage timers of its run."""
        if multi:
            comm = comms[h._slot.ctx.value]
            tg1 = time.perf_counter()
            # enqueued behind the plan's run on the context's communication stream (checked return code); the context
            # takes its next ensemble at once, the transfers are waited for before the clock stops
            comm.gather_plan(h.plan, sizes, root=0, host=False)
            gather_t[1] += time.perf_counter() - tg1
            gather_t[2] += 1
        if timed[0]:
            for name, v in h.plan.stage_ms().items():      # HIP events of the critical lane (waits for the plan)
                stage_acc[name] = stage_acc.get(name, 0.0) + v
            nstage[0] += 1

    def sync_ranks():
        eng.drain()
        for c in comms.values():
            c.wait()                          # every gather of the timed region has landed
        if rdzv is not None:
            rdzv.barrier()

    # warm-up: one ensemble at a time, which also gives the single-ensemble latency; every context runs once
    latency = None
    for s in range(max(args.warmup, nfl)):
        tw = run_api_loop(eng, [works[s % nfl]], 1, 1, resident=True, on_done=on_done)
        latency = tw if latency is None else min(latency, tw)
    run_api_loop(eng, works, nfl, nfl, resident=True, on_done=on_done)      # every context holds its plan + signals
    sync_ranks()
    timed[0] = True
    t0 = time.perf_counter()
    run_api_loop(eng, works, args.steps, nfl, resident=True, on_done=on_done)
    sync_ranks()
    elapsed = time.perf_counter() - t0
    if rdzv is not None:
        elapsed = rdzv.max(elapsed)
    timed[0] = False

    # ---- verification outside the timed region: every member converged; (multi) the gathered blocks are the results
    last = eng.submit(*works[0][:3], works[0][2], p=1, q=0.0, dwell=DWELL, resident=True)
    ref = last.result(check=False)
    ok = int((ref.status == 0).sum())
    n0 = last.plan.lane0_members()        # members (the largest) whose stage timers `stage_ms` reports
    nfb = last.plan.eig_fallbacks()       # members the Ehrlich-Aberth eigenvalue path handed to the QR iteration
    if multi:
        buf = comms[last._slot.ctx.value].gather_plan(last.plan, sizes, root=-1)       # to every rank, for the check
        off = int(sizes[:rank].sum())
        ll, sv, st, kp = unpack_block(buf[off:off + int(sizes[rank])], int(ms.sum()), int(ms.sum()), len(ms))
        gathered_ok[0] = bool(np.array_equal(ll, ref.lines) and np.array_equal(sv, ref.sv) and
                              np.array_equal(st, ref.status) and np.array_equal(kp, ref.keep.astype(bool)))
        flags = rdzv.allgather(bytes([1 if gathered_ok[0] else 0]) + int(ok).to_bytes(4, "little"))
        gathered_ok[0] = all(f[0] == 1 for f in flags)
        ok_all = [int.from_bytes(f[1:5], "little") for f in flags]
    else:
        ok_all = [ok]

    # ---- the same steps one at a time, host -> host, and the other configurations (N = 1 only)
    serial = host_incl = None
    others = {}
    if not multi:
        ns_ser = min(args.steps, 5)
        if nfl > 1:
            ser_acc, ser_n = {}, [0]

            def ser_done(h):
                for name, v in h.plan.stage_ms().items():
                    ser_acc[name] = ser_acc.get(name, 0.0) + v
                ser_n[0] += 1
            ts = run_api_loop(eng, [works[0]], ns_ser, 1, resident=True, on_done=ser_done)
            serial = {"value": units * ns_ser / ts, "ms_per_step": 1e3 * ts / ns_ser, "steps": ns_ser, "ensembles_in_flight": 1,
                      "stage_ms": {k: v / max(1, ser_n[0]) for k, v in ser_acc.items()}}
        if not args.no_extras:
            # SURVEY.md 8d's wording of the metric: host signals resident -> host line lists resident, through the same
            # public API, the same number of ensembles in flight
            nh = args.steps
            th = run_api_loop(eng, works, nh, nfl, resident=False)
            th1 = run_api_loop(eng, [works[0]], ns_ser, 1, resident=False)
            host_incl = {"value": units * nh / th, "unit": "solves/s", "ms_per_step": 1e3 * th / nh, "steps": nh,
                         "ensembles_in_flight": nfl, "one_ensemble_at_a_time": units * ns_ser / th1,
                         "includes": "Engine.submit -> Pending.result: staging + H2D of the signal, every kernel, D2H of "
                                     "lines, sv, mu, keep, status"}
            if args.workload == "C2":
                eng.clear_plan_cache()
                for name, steps_o, fl_o in (("NS", 8 * nfl, nfl), ("C3", 2, 1), ("C5", 1, 1), ("C4", 1, 1)):
                    try:
                        w = [make_workload(name, 1000 * k) for k in range(fl_o)]
                        run_api_loop(eng, w, fl_o, fl_o, resident=False)                     # plans + warm-up
                        to = run_api_loop(eng, w, steps_o, fl_o, resident=False)
                        chk = eng.submit(*w[0][:3], w[0][2], p=1, q=0.0, dwell=DWELL).result(check=False)
                        hk = None
                        if name == "C3":
                            # the Hankel build on a launch that fills the chip (the C2 launches are 35-55 us: launch-latency
                            # bound): lane 0's members of this run, algorithmic bytes 16 m^2 written + 16 (2m - 1) read per member
                            h = eng.submit(*w[0][:3], w[0][2], p=1, q=0.0, dwell=DWELL)
                            h.result(check=False)
                            n0h = h.plan.lane0_members()
                            mm = sorted((int(x) for x in w[0][2]), reverse=True)[:n0h]
                            hb = sum(16.0 * x * x + 16.0 * (2 * x - 1) for x in mm)
                            ms_h = h.plan.stage_ms()["k_hankel"]
                            hk = {"kernel": "k_hankel", "members": n0h, "bytes": hb, "ms": ms_h, "GBps": hb / (ms_h * 1e-3) / 1e9,
                                  "frac_of_hbm_peak": hb / (ms_h * 1e-3) / 1e9 / HBM_PEAK_GBS}
                        others[name] = {"workload": WORKLOADS[name], "value": len(w[0][2]) * steps_o / to, "unit": "solves/s", "hankel_build": hk,
                                        "ms_per_step": 1e3 * to / steps_o, "steps": steps_o, "ensembles_in_flight": fl_o,
                                        "members": int(len(w[0][2])), "members_ok": int((chk.status == 0).sum()),
                                        "eig_fallbacks": eng._slots[0].plans and max(pl.eig_fallbacks() for pl in eng._slots[0].plans.values()),
                                        "timed": "host -> host through Engine.submit"}
                    except Exception as e:       # informational lines: never lose the headline over them
                        others[name] = {"error": repr(e)}
                    eng.clear_plan_cache()

    if rank == 0:
        value = (1 if args.sharded else world) * units * args.steps / elapsed
        stage_ms = {k: v / max(1, nstage[0]) for k, v in stage_acc.items()}
        # stage timers are those of lane 0 (the largest members, the critical path): price its launches
        # with the flops of exactly those members; `pipeline_tflops` below uses all members
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
        kname = {"k_hqr": "k_ab_iter" if os.environ.get("KBDM_EIG_AB", "1") != "0" else "k_hqr2_team",
                 "k_svd_fac": "k_bidiag_panel<0>", "k_hess": "k_hess_panel",
                 "k_gen(Q,P)": "k_wy_update", "k_dc_final": "k_dc_final", "k_invit": "k_invit_reg<8>"}.get(dom, dom)
        # HBM bytes per launch of that kernel: NOT measured by this run - taken from the newest committed PMC passes
        # (profiles/*_pmc_traffic.json, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this same command), if any
        traffic = traffic_src = None
        try:
            import glob
            def newest_first(fn):          # r<round>_<milestone>_pmc_traffic.json: later round, then fin > end / last > mid > others
                tag = os.path.basename(fn).split("_")
                return (tag[0], {"fin": 3, "end": 2, "last": 2, "mid": 1}.get(tag[1], 0))
            for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), key=newest_first, reverse=True):
                pmc = json.load(open(fn))["kernels"]
                if kname in pmc:
                    traffic = pmc[kname].get("hbm_bytes_per_launch")
                    traffic_src = "profiles/" + os.path.basename(fn) + " (committed rocprofv3 PMC passes, not this run)"
                    break
        except Exception:
            traffic = None
        roofline = {"kernel": kname, "stage_timer": dom,
                    "bound": "mfma" if kname in ("k_trail_update", "k_hess_update", "k_ab_iter") else "fp64_vector",
                    "bound_note": ("eigenvalue stage: Ehrlich-Aberth iterations (k_ab_iter: FP64-MFMA block products + a serial "
                                   "32-row triangle per block), priced with the stage's algorithmic 65.33 l^3 flops per member "
                                   "(SURVEY.md 8d) over the stage's HIP-event time" if kname == "k_ab_iter" else
                                   "FP64 vector FMA, instruction-issue / latency bound; MI355X FP64 vector peak = FP64 matrix "
                                   "peak = 78.6 TFLOP/s"),
                    "achieved": achieved, "peak": FP64_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                    "avg_ms": stage_ms[dom], "algorithmic_flops_per_launch": fl[dom],
                    "launch_members": n0}
        if kname == "k_ab_iter":
            roofline["launches_per_stage"] = "k_ab_leaf + 24 x levels k_ab_iter + k_ab_finish; `traffic` is ONE k_ab_iter launch"
        if serial and serial.get("stage_ms", {}).get(dom, 0) > 0:
            # the same stage with the GPU to itself (in the timed region four ensembles share the CUs, so a stage's HIP-event
            # time there is its share of a busy chip, not its cost)
            alone = fl[dom] / (serial["stage_ms"][dom] * 1e-3) / 1e12
            roofline["one_ensemble_at_a_time"] = {"avg_ms": serial["stage_ms"][dom], "achieved": alone, "frac": alone / FP64_PEAK_TFLOPS}
        total_fl = sum(fl_all.values())
        pipe = total_fl * args.steps * (1 if args.sharded else world) / elapsed / 1e12
        roofline["whole_pipeline"] = {"achieved": pipe, "frac": pipe / (FP64_PEAK_TFLOPS * world), "unit": "TFLOP/s",
                                      "note": "algorithmic flops of every stage of every member (SURVEY.md 8d) over the timed region"}
        out = {
            "metric": "KBDM solves/sec over m-range ensemble, N=2048 complex signal",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong" if args.sharded else "weak",
            "vs_baseline": None, "dtype": "f64 (complex128)", "data": "synthetic",
            "config": {"workload": wname, "members_per_gpu": len(ms), "members_per_step": units * (1 if args.sharded else world),
                       "parallelism": (f"one job sharded over {world} rank(s) by LPT on m^3" if args.sharded
                                       else f"one ensemble per rank x{world}"),
                       "ensembles_in_flight": nfl,
                       "api": "llckbdm_amd.engine.Engine.submit / Pending.result (the product's scheduler)",
                       "timed_region": "signals resident in HBM -> results in host memory (host -> host: `host_to_host`)",
                       "collective": ("one grouped RCCL send/recv of the packed results to rank 0 per step (kbdm_plan_gather)"
                                      if multi else "none")},
            "roofline": roofline,
            "pipeline_tflops": total_fl * args.steps * (1 if args.sharded else world) / elapsed / 1e12,
            "stage_ms": stage_ms,
            "step_latency_ms": None if latency is None else 1e3 * latency,
            "one_ensemble_at_a_time": serial,
            "host_to_host": host_incl,
            "other_configs": others or None,
            "eig_fallbacks_last_step": nfb,
            "members_ok": min(ok_all), "members_ok_per_rank": ok_all,
            "gathered_blocks_verified": gathered_ok[0] if multi else None,
            "gather_host_ms_per_call": 1e3 * gather_t[1] / max(1, gather_t[2]) if multi else None,
            "git_head": git_head(), "bench_sha256_16": file_sha(os.path.abspath(__file__)),
            "lib_sha256_16": file_sha(_lib.LIB_PATH),
        }
        if world == 1 and not multi and not args.no_cpu_baseline:
            try:
                if args.workload in ("C2", "NS"):
                    sig0 = make_workload(args.workload, rank)[0][0]
                    out["cpu_baseline"] = cpu_baseline(sig0, ms, DWELL)
                    out["cpu_baseline"]["serial"] = cpu_baseline_serial(sig0, ms, DWELL)
                else:
                    out["cpu_baseline"] = None
            except Exception as e:   # the baseline is informational; never lose the GPU number over it
                out["cpu_baseline"] = {"error": repr(e)}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    for c in reversed(list(comms.values())):      # borrowers detach first, the owner destroys
        c.close()
    if rdzv is not None:
        rdzv.barrier()
        rdzv.close()
    eng.close()


if __name__ == "__main__":
    main()
