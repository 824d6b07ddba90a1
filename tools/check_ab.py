"""GPU: the Ehrlich-Aberth eigenvalue path against the QR iteration on the bench workloads: fallbacks, eigenvalue
differences, kept-line differences, time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from llckbdm_amd import datasets
from llckbdm_amd.engine import Engine
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
work = {"C2": lambda: datasets.config2(seed=0), "NS": lambda: datasets.north_star(seed=0),
        "C3": lambda: datasets.config3(count=128, m=512), "C4": lambda: tuple(x[::8] if i else x for i, x in enumerate(datasets.config4())),
        "C1": lambda: (datasets.brain_sim_signal(1024).reshape(1, -1), np.zeros(3, np.int32), np.array([300, 200, 150], np.int32))}[name]()
sigs, sidx, ms = work
res = {}
for ab in ("1", "0"):
    os.environ["KBDM_EIG_AB"] = ab
    eng = Engine(0, in_flight=1)
    plan = eng.plan(sigs.shape[0], sigs.shape[1], sidx, ms, ms, dwell=datasets.DWELL)
    plan.upload(sigs)
    plan.execute(); plan.execute()
    plan.ab_stats()
    t = time.perf_counter(); plan.execute(); dt = time.perf_counter() - t
    st = plan.ab_stats()
    if ab == "1":
        for srow in st[:6]:
            if srow.any(): print("   tiles working per iteration:", srow.tolist())
    r = plan.download()
    print(f"EIG_AB={ab}: {1e3*dt:8.2f} ms  fallbacks {plan.eig_fallbacks()} of {len(ms)}  status!=0: {int((r.status != 0).sum())}  stage {dict((k, round(v, 2)) for k, v in plan.stage_ms().items() if v > 0.3)}", flush=True)
    res[ab] = r
    eng.close()
a, b = res["1"], res["0"]
worst = 0.0
for i in range(len(ms)):
    za, zb = a.eigenvalues(i), b.eigenvalues(i)
    d = np.abs(za[:, None] - zb[None, :])
    worst = max(worst, d.min(axis=1).max(), d.min(axis=0).max())
    ka, kb = a.line_list(i)[a.keep_mask(i)], b.line_list(i)[b.keep_mask(i)]
    if len(ka) != len(kb):
        print("member", i, "m", ms[i], "kept", len(ka), len(kb))
print("max eigenvalue mismatch (nearest-neighbour, both ways):", worst)
print("sv equal:", np.array_equal(a.sv, b.sv))
