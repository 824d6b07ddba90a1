"""Probe: C2 ensembles in flight at once on one GPU (one Engine = one set of streams each)."""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("PROBE_HWQ", "8"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llckbdm_amd import datasets
from llckbdm_amd.engine import Engine

nfl = int(os.environ.get("PROBE_INFLIGHT", "2"))
steps = int(os.environ.get("PROBE_STEPS", "6"))
engs, plans = [], []
for k in range(nfl):
    sigs, sig_idx, ms = datasets.config2(seed=k)
    e = Engine(0)
    p = e.plan(sigs.shape[0], sigs.shape[1], sig_idx, ms, ms, p=1, q=0.0, dwell=datasets.DWELL)
    p.upload(sigs)
    engs.append(e); plans.append(p)
for p in plans:
    p.execute(sync=True)
# sequential on plan 0
t0 = time.perf_counter()
for _ in range(steps):
    plans[0].execute(sync=True)
seq = (time.perf_counter() - t0) / steps
# nfl in flight: submit round-robin, wait for a plan only when it is needed again
t0 = time.perf_counter()
for s in range(steps):
    p = plans[s % nfl]
    p.sync()
    p.execute(sync=False)
for p in plans:
    p.sync()
ovl = (time.perf_counter() - t0) / steps
ok = [int((p.download().status == 0).sum()) for p in plans]
print(json.dumps({"inflight": nfl, "lanes": os.environ.get("KBDM_LANES"), "seq_ms": 1e3 * seq, "overlapped_ms_per_step": 1e3 * ovl,
                  "solves_per_s": 151 / ovl, "ok": ok}), flush=True)
