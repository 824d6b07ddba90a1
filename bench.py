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
  roofline     : the dominant KERNEL of a clean pass (one ensemble at a time, the GPU to itself): its algorithmic bytes or FP64
                 flops per launch / its average launch duration, from HIP events the library records around every launch on the
                 stream the kernel runs on (KBDM_MODE_KERNEL_TIMERS); + `top_kernels`, `top_stages`, `north_star` (MFMA
                 utilisation of the SVD panel update and of the Hessenberg update over all launches, Hankel GB/s) and
                 `whole_pipeline` (216 m^3 per member, SURVEY.md 8d, over the timed region)
  value_host_to_host / host_to_host : the same loop with the upload of the signals inside it (SURVEY.md 8d's unit)
  one_ensemble_at_a_time, sample_kbdm_call, llc_kbdm_c2 : ONE ensemble at a time on the engine's context for synchronous
                 calls (panel teams), the same through llckbdm_amd.sampling.sample_kbdm, and llc_kbdm end to end
  cpu_baseline : the numpy/scipy oracle (the reference's own LAPACK calls) timed on the host cores of this box on a
                 bounded sample of the same workload (rank 0, N=1 only): one process per core on EVERY core of the box (and,
                 as `host_share_16_cores`, on the 16-core share of one GPU), and, as `serial`, the shape the reference ships
                 (sampling.py:52-62: one loop, default BLAS threads); the CPU model is stated.
  other_configs: short runs of the other BASELINE.json configurations (C3, C4, C5) and of the north-star ensemble
                 m = 100..500 through the same public API (N=1 only).
--trace-mode (rocprofv3 runs): every ensemble of the process runs in flight, nothing else runs.
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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return None


def cpu_baseline(sig, ms, dwell, min_seconds=10.0, max_passes=8, cores=None):
    """Oracle (numpy/scipy: the reference's own LAPACK calls) over the WHOLE C2 member list,
    one process per host core with 1 BLAS thread each, repeated until >= min_seconds of wall.
    cores=None: every core this process may run on (len(os.sched_getaffinity(0)))."""
    import multiprocessing as mp
    from threadpoolctl import threadpool_limits
    members = [int(m) for m in ms][::-1]            # longest first: better balance
    avail = max(1, min(len(os.sched_getaffinity(0)), (os.cpu_count() or 1)))
    cores = max(1, min(len(members), avail if cores is None else min(cores, avail)))
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
    return {"value": passes * len(members) / dt, "unit": "solves/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
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


def run_api_loop(eng, works, steps, nfl, resident, on_done=None, wide=False):
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
        pend.append(eng.submit(sg, si, mm, mm, p=1, q=0.0, dwell=DWELL, resident=resident, _wide=wide and nfl == 1))
    while pend:
        h = pend.popleft()
        if on_done:
            on_done(h)
        h.result(check=False)
    return time.perf_counter() - t0


DWELL = 5e-4


def hankel_alone(work, members=512):
    """The Hankel build with the GPU to itself: a context with ONE lane (in the pipeline the lanes' Hankel launches overlap and
    share the HBM: round 3's figure was one lane's half), the first `members` members of the C3 workload, HIP events around
    the k_hankel launch on its stream.  Algorithmic bytes: 16 m^2 written + 16 (2m - 1) read per member."""
    from llckbdm_amd.engine import Engine
    sig, idx, m = work[0], work[1][:members], work[2][:members]
    old = {k: os.environ.get(k) for k in ("KBDM_LANES", "KBDM_WIDE_SOLVE")}
    os.environ["KBDM_LANES"], os.environ["KBDM_WIDE_SOLVE"] = "1", "0"
    try:
        e1 = Engine(0, in_flight=1)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    rank64 = None
    try:
        best = None
        for _ in range(3):
            h = e1.submit(sig, idx, m, m, p=1, q=0.0, dwell=DWELL)
            h.result(check=False)
            assert h.plan.lane0_members() == len(m)
            ms_h = h.plan.stage_ms()["k_hankel"]
            best = ms_h if best is None else min(best, ms_h)
        try:
            # north_star's "MFMA utilisation in the SVD panel update" on launches that fill the chip, with nothing else on the
            # GPU (with two lanes the other lane's panel workgroups sit in front of an update's tiles and its events time them)
            cst, ckm = clean_profile(e1, (sig, idx, m), reps=1)
            ck, _ = build_rooflines(m, len(m), cst, ckm)
            rank64 = {r["kernel"]: {"frac_of_fp64_matrix_peak": r["frac"], "TFLOPs": r["achieved"], "avg_ms": r["avg_ms"],
                                    "launches": r["launches"], "members_per_launch": int(len(m))}
                      for r in ck if r["kernel"] in ("k_trail_update", "k_hess_update")}
        except Exception as e:
            rank64 = {"error": repr(e)}
    finally:
        e1.close()
    hb = sum(16.0 * int(x) * int(x) + 16.0 * (2 * int(x) - 1) for x in m)
    return {"kernel": "k_hankel", "members": int(len(m)), "bytes": hb, "ms": best, "GBps": hb / (best * 1e-3) / 1e9,
            "frac_of_hbm_peak": hb / (best * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "measured": "one lane, nothing else on the GPU; best of 3 launches (HIP events on the launch's stream); a write-only "
                        "stream: the same kernel alone under rocprofv3 (tools/hankel_bw.sh) reads 4.27 TB/s"}, rank64


def pmc_c3():
    """MFMA busy fraction of the rank-64 updates on C3 from the newest committed PMC pass (not measured by this run)."""
    import glob
    fns = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_north_star_kernels_c3.json")))
    if not fns:
        return None
    d = json.load(open(fns[-1]))
    return {"k_trail_update": d["k_trail_update"].get("pmc_mfma_utilisation"), "k_hess_update": d["k_hess_update"].get("pmc_mfma_utilisation"),
            "source": "profiles/" + os.path.basename(fns[-1])}


def clean_profile(eng, work, reps=3):
    """One ensemble at a time, the GPU to itself, with the library's per-kernel HIP-event timers on (KBDM_MODE_KERNEL_TIMERS:
    events on the stream each kernel runs on, around every launch of the timed kernel classes): the basis of the roofline
    objects.  Returns (stage_ms, {kernel: (total ms, launches)}) averaged over `reps` runs, for lane 0's members."""
    from llckbdm_amd import _lib
    sg, si, mm = work
    eng.drain()
    sig = np.ascontiguousarray(np.atleast_2d(sg), dtype=np.complex128)
    plan = eng.cached_plan(sig.shape[0], sig.shape[1], si, mm, mm, 1, 0.0, DWELL)
    plan.set_mode(_lib.MODE_KERNEL_TIMERS)
    st_acc, k_acc = {}, {}
    try:
        for _ in range(reps + 1):
            plan.submit(sig)
            plan.collect()
            st, km = plan.stage_ms(), plan.kernel_ms()
            if _ == 0:
                continue                       # (the first run creates the events)
            for k, v in st.items():
                st_acc[k] = st_acc.get(k, 0.0) + v / reps
            for k, (ms_, n_) in km.items():
                a = k_acc.get(k, (0.0, 0))
                k_acc[k] = (a[0] + ms_ / reps, n_)
    finally:
        plan.set_mode(0)
    return st_acc, k_acc


def build_rooflines(ms, n0, stage_ms, kernel_ms):
    """Roofline entries from the clean pass.  Kernels with timers of their own are priced per launch (algorithmic flops or
    bytes of lane 0's members per launch / average launch duration); the remaining stages by their stage timer."""
    lane0 = sorted((int(m) for m in ms), reverse=True)[:n0]
    fl = {}
    for m in lane0:
        for k, v in stage_flops(m, m).items():
            fl[k] = fl.get(k, 0.0) + v
    npan = lambda m: (m - 64) // 32 if m >= 96 else 0
    # the one-stage panels stream the trailing matrix twice (bidiagonalisation) / once (Hessenberg: the rows below the panel's first x the trailing
    # columns) per column: the bytes of the ALGORITHM, not compulsory traffic
    b_bidiag = sum(sum(2 * 16.0 * (m - c) ** 2 for c in range(npan(m) * 32)) for m in lane0)
    b_hess = sum(sum(16.0 * (m - (c // 32) * 32 - 1) * (m - c - 1) for c in range(npan(m) * 32)) for m in lane0)
    b_hankel = sum(16.0 * m * m + 16.0 * (2 * m - 1) for m in lane0)
    # flops of the blocked reductions inside their panel columns: half in the panel's matrix-vector products, half in the update
    f_trail = sum(sum(8.0 * 64 * (m - 32 * (p + 1)) ** 2 for p in range(npan(m))) for m in lane0)     # rank-64 complex updates
    f_hupd = sum(sum(8.0 * 64 * m * (m - 32 * (p + 1)) for p in range(npan(m))) for m in lane0)
    spec = {   # kernel: (bound, algorithmic quantity over all launches of the stage, unit, note)
        "k_ab_iter": ("mfma", fl["k_hqr"], "flop", "eigenvalue stage (Ehrlich-Aberth): 65.33 l^3 per member (SURVEY 8d)"),
        "k_bidiag_panel_team": ("hbm", b_bidiag, "B", "two passes over the trailing matrix per panel column, 32 (n - j)^2 B"),
        "k_hess_panel_team": ("hbm", b_hess, "B", "one pass over the rows below the panel's first row x the trailing columns per panel column, 16 (n - p0 - 1)(n - k - 1) B"),
        "k_trail_update": ("mfma", f_trail, "flop", "rank-64 trailing update: the zgemm half of the blocked bidiagonalisation (north_star's SVD panel update)"),
        "k_hess_update": ("mfma", f_hupd, "flop", "rank-64 update of the Hessenberg reduction"),
        "k_hankel": ("hbm", b_hankel, "B", "16 m^2 B written + the signal segment read per member (pipeline: U^{p-1} only)"),
    }
    out = []
    for kname, (bound, qty, unit, note) in spec.items():
        tot, n = kernel_ms.get(kname, (0.0, 0))
        if n <= 0 or tot <= 0:
            continue
        avg_ms = tot / n
        per_launch = qty / n
        if bound == "mfma":
            ach = per_launch / (avg_ms * 1e-3) / 1e12
            peak, u = FP64_PEAK_TFLOPS, "TFLOP/s"
        else:
            ach = per_launch / (avg_ms * 1e-3) / 1e9
            peak, u = HBM_PEAK_GBS, "GB/s"
        out.append({"kernel": kname, "bound": bound, "achieved": ach, "peak": peak, "unit": u, "frac": ach / peak,
                    "avg_ms": avg_ms, "launches": n, "total_ms": tot,
                    ("algorithmic_flops_per_launch" if unit == "flop" else "algorithmic_bytes_per_launch"): per_launch,
                    "launch_members": n0, "note": note})
    out.sort(key=lambda r: -r["total_ms"])
    stages = []
    for k, v in sorted(stage_ms.items(), key=lambda kv: -kv[1]):
        if v <= 0:
            continue
        a = fl.get(k, 0.0) / (v * 1e-3) / 1e12
        stages.append({"stage": k, "ms": v, "algorithmic_flops": fl.get(k, 0.0), "achieved_tflops": a, "frac_fp64": a / FP64_PEAK_TFLOPS})
    return out, stages[:6]


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
    ap.add_argument("--trace-mode", action="store_true",
                    help="for rocprofv3 runs: every ensemble of the process runs with --in-flight ensembles in flight (warm-up "
                         "included), nothing else runs (no one-at-a-time pass, no clean pass, no sampler call, no extras, no CPU baseline)")
    ap.add_argument("--no-stagger", dest="stagger", action="store_false",
                    help="submit the ensembles of a burst at once instead of a fraction of a cycle apart")
    args = ap.parse_args()
    if args.trace_mode:
        args.no_extras = args.no_cpu_baseline = True
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
    if not args.trace_mode:
        for s in range(max(args.warmup, nfl)):
            tw = run_api_loop(eng, [works[s % nfl]], 1, 1, resident=True, on_done=on_done)
            latency = tw if latency is None else min(latency, tw)
    run_api_loop(eng, works, max(nfl, args.warmup if args.trace_mode else 0), nfl, resident=True, on_done=on_done)      # every context holds its plan + signals
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
    if args.trace_mode and not multi:     # (no lone ensemble in a trace that is meant to be all in flight)
        pl0 = next(iter(eng._slots[0].plans.values()))
        ok, n0, nfb, last, ref = len(ms), pl0.lane0_members(), pl0.eig_fallbacks(), None, None
    else:
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
    serial = host_incl = clean = sample_call = llc_call = None
    others = {}
    if not multi and not args.trace_mode:
        ns_ser = min(args.steps, 5)
        if nfl > 1:
            ser_acc, ser_n = {}, [0]

            def ser_done(h):
                for name, v in h.plan.stage_ms().items():
                    ser_acc[name] = ser_acc.get(name, 0.0) + v
                ser_n[0] += 1
            run_api_loop(eng, [works[0]], 2, 1, resident=True, wide=True)              # (plan + warm-up of that context)
            ts = run_api_loop(eng, [works[0]], ns_ser, 1, resident=True, on_done=ser_done, wide=True)
            serial = {"value": units * ns_ser / ts, "ms_per_step": 1e3 * ts / ns_ser, "steps": ns_ser, "ensembles_in_flight": 1,
                      "context": "the engine's context for synchronous calls (panel teams: Engine.WIDE_*), as Engine.solve uses it",
                      "stage_ms": {k: v / max(1, ser_n[0]) for k, v in ser_acc.items()}}
        # the clean pass behind the roofline objects: one ensemble at a time with the per-kernel timers on
        clean = None
        try:
            clean = clean_profile(eng, works[0])
        except Exception as e:
            clean = None
            print("clean_profile failed:", repr(e), file=sys.stderr)
        sample_call = llc_call = None
        if args.workload in ("C2", "NS"):
            # ONE call of the drop-in sampler (what llc_kbdm makes: llckbdm.py:76-83), host signal in, python line lists out
            from llckbdm_amd.sampling import sample_kbdm
            sg0 = works[0][0][0]
            mr = [int(x) for x in ms]
            sample_kbdm(sg0, DWELL, mr, p=1, l=None, q=0, engine=eng)
            tsc = time.perf_counter()
            nrep = 3
            for _ in range(nrep):
                sample_kbdm(sg0, DWELL, mr, p=1, l=None, q=0, engine=eng)
            tsc = (time.perf_counter() - tsc) / nrep
            sample_call = {"value": len(mr) / tsc, "unit": "solves/s", "ms_per_call": 1e3 * tsc,
                           "api": "llckbdm_amd.sampling.sample_kbdm(data, dwell, m_range, p, l, q): one synchronous call, host -> host"}
        if not args.no_extras:
            if args.workload == "C2":
                try:
                    from llckbdm_amd.llckbdm import llc_kbdm
                    tl = time.perf_counter()
                    res_llc = llc_kbdm(works[0][0][0], DWELL, [int(x) for x in ms], p=1, l=None, q=0.0, engine=eng)
                    llc_call = {"seconds": time.perf_counter() - tl, "lines": int(len(res_llc.line_list)),
                                "api": "llckbdm_amd.llckbdm.llc_kbdm on the C2 ensemble: sampler + 150-fit clustering sweep + scoring"}
                except Exception as e:
                    llc_call = {"error": repr(e)}
            # SURVEY.md 8d's wording of the metric: host signals resident -> host line lists resident, through the same
            # public API, the same number of ensembles in flight
            nh = args.steps
            th = run_api_loop(eng, works, nh, nfl, resident=False)
            th1 = run_api_loop(eng, [works[0]], ns_ser, 1, resident=False, wide=True)
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
                            hk, c3_mfma = hankel_alone(w[0])
                        others[name] = {"workload": WORKLOADS[name], "value": len(w[0][2]) * steps_o / to, "unit": "solves/s", "hankel_build": hk,
                                        "ms_per_step": 1e3 * to / steps_o, "steps": steps_o, "ensembles_in_flight": fl_o,
                                        "members": int(len(w[0][2])), "members_ok": int((chk.status == 0).sum()),
                                        "eig_fallbacks": eng._slots[0].plans and max(pl.eig_fallbacks() for pl in eng._slots[0].plans.values()),
                                        "timed": "host -> host through Engine.submit"}
                        if name == "C3":
                            others[name]["rank64_updates"] = c3_mfma
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
        total_fl = sum(fl_all.values())
        nj = (1 if args.sharded else world)
        pipe = total_fl * args.steps * nj / elapsed / 1e12
        pipe216 = 216.0 * float(np.sum(np.asarray(ms, dtype=np.float64) ** 3)) * args.steps * nj / elapsed / 1e12
        roofline = None
        if clean is not None:
            kern, stages = build_rooflines(ms, n0, clean[0], clean[1])
            if kern:
                roofline = dict(kern[0])
                roofline["measured"] = ("one ensemble at a time, the GPU to itself: HIP events recorded by the library on the "
                                        "stream the kernel runs on, around every launch (KBDM_MODE_KERNEL_TIMERS)")
                # HBM bytes per launch of that kernel from the newest committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
                # WRITE_SIZE, gfx950 correction applied: profiles/*_pmc_traffic.json) - not measured by this run
                traffic = traffic_src = None
                try:
                    import glob
                    def newest_first(fn):
                        tag = os.path.basename(fn).split("_")
                        return (tag[0], {"fin": 3, "end": 2, "last": 2, "mid": 1}.get(tag[1], 0))
                    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), key=newest_first, reverse=True):
                        pmc = json.load(open(fn))["kernels"]
                        hit = [k for k in pmc if k.split("<")[0] == roofline["kernel"].split("<")[0]]
                        if hit:
                            traffic = pmc[hit[0]].get("hbm_bytes_per_launch")
                            traffic_src = "profiles/" + os.path.basename(fn) + " (committed rocprofv3 PMC passes, not this run)"
                            break
                except Exception:
                    traffic = None
                roofline["traffic"] = traffic
                roofline["traffic_source"] = traffic_src
                roofline["top_kernels"] = kern[:5]
                roofline["top_stages"] = stages
                tu = next((r for r in kern if r["kernel"] == "k_trail_update"), None)
                hu = next((r for r in kern if r["kernel"] == "k_hess_update"), None)
                hk2 = next((r for r in kern if r["kernel"] == "k_hankel"), None)
                roofline["north_star"] = {
                    "mfma_util_svd_panel_update": None if tu is None else tu["frac"],
                    "mfma_util_hess_update": None if hu is None else hu["frac"],
                    "hankel_build_GBps_c2_launch": None if hk2 is None else hk2["achieved"],
                    "mfma_util_svd_panel_update_c3": ((others.get("C3") or {}).get("rank64_updates") or {}).get("k_trail_update"),
                    "mfma_util_hess_update_c3": ((others.get("C3") or {}).get("rank64_updates") or {}).get("k_hess_update"),
                    "pmc_c3": pmc_c3(),
                    "note": "MFMA utilisation = algorithmic flops of the rank-64 update launches / their HIP-event time / 78.6 "
                            "TFLOP/s, over ALL launches of lane 0 in the clean pass: C2's launches cover 32 members (sub-chip: "
                            "the last panels launch 128 workgroups); the _c3 figures: 512 members of C3 on a one-lane context "
                            "(launches that fill the chip, nothing else on the GPU); pmc_c3: "
                            "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs) from the committed rocprofv3 pass "
                            "(tools/mfma_c3.sh); the Hankel build on a chip-filling launch: other_configs.C3.hankel_build"}
        if roofline is None:
            roofline = {"kernel": None, "bound": "mfma", "achieved": None, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None,
                        "traffic": None, "note": "no clean pass in this mode (N > 1): see the N = 1 line"}
        roofline["whole_pipeline"] = {"achieved": pipe216, "frac": pipe216 / (FP64_PEAK_TFLOPS * world), "unit": "TFLOP/s",
                                      "flops_per_member": "216 m^3 (SURVEY.md 8d: F_alg at l = m) over the timed region",
                                      "by_stage_model": {"achieved": pipe, "note": "sum of the per-stage algorithmic counts this "
                                                         "file's stage_flops() prices the stages with (167 m^3)"}}
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
                       "timed_region": "signals resident in HBM when the clock starts -> results in host memory (the bench "
                                       "contract); SURVEY 8d's host -> host unit is `value_host_to_host` / `host_to_host`",
                       "collective": ("one grouped RCCL send/recv of the packed results to rank 0 per step (kbdm_plan_gather)"
                                      if multi else "none")},
            "roofline": roofline,
            "value_host_to_host": None if host_incl is None else host_incl["value"],
            "sample_kbdm_call": sample_call,
            "llc_kbdm_c2": llc_call,
            "pipeline_tflops": pipe216,
            "stage_ms": stage_ms,
            "step_latency_ms": None if latency is None else 1e3 * latency,
            "one_ensemble_at_a_time": serial,
            "host_to_host": host_incl,
            "other_configs": others or None,
            "eig_fallbacks_last_step": nfb,
            "members_ok": min(ok_all), "members_ok_per_rank": ok_all,
            "gathered_blocks_verified": gathered_ok[0] if multi else None,
            "rccl_world_seen": (sorted(set(int(c.world) for c in comms.values())) if multi else None),
            "gather_host_ms_per_call": 1e3 * gather_t[1] / max(1, gather_t[2]) if multi else None,
            "git_head": git_head(), "bench_sha256_16": file_sha(os.path.abspath(__file__)),
            "lib_sha256_16": file_sha(_lib.LIB_PATH),
        }
        if world == 1 and not multi and not args.no_cpu_baseline:
            try:
                if args.workload in ("C2", "NS"):
                    sig0 = make_workload(args.workload, rank)[0][0]
                    out["cpu_baseline"] = cpu_baseline(sig0, ms, DWELL)                       # every core of the box
                    if out["cpu_baseline"]["cores"] > 16:                                      # and the 16-core host share of one GPU
                        out["cpu_baseline"]["host_share_16_cores"] = cpu_baseline(sig0, ms, DWELL, min_seconds=6.0, cores=16)
                    out["cpu_baseline"]["serial"] = cpu_baseline_serial(sig0, ms, DWELL, stride=8)
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
