"""GPU: wall-clock split of a full k_ab_iter tile (KBDM_AB_DBG=8 instrumentation) on the C2 ensemble."""
import os, sys
os.environ["KBDM_AB_DBG"] = str(8 | int(sys.argv[1]) if len(sys.argv) > 1 else 8)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from llckbdm_amd import datasets
from llckbdm_amd.engine import Engine
sigs, sidx, ms = datasets.config2(seed=0)
eng = Engine(0, in_flight=1)
plan = eng.plan(sigs.shape[0], sigs.shape[1], sidx, ms, ms, dwell=datasets.DWELL)
plan.upload(sigs)
plan.execute(); plan.ab_stats()
plan.execute()
st = plan.ab_stats()
row = st[8].astype(np.int64)
cnt = max(1, int(row[23]))
names = ["prologue", "H triangle + reciprocals", "block product (chunks)", "accumulators -> LDS", "triangle + row stores", "rescale / loop end"]
tot = row[:6].sum()
print("full tiles of nodes > 256 rows:", cnt, " mean tile", round(tot * 0.01 / cnt, 1), "us")
for i, nm in enumerate(names):
    print(f"  {nm:28s} {row[i] * 0.01 / cnt:8.1f} us  {100.0 * row[i] / tot:5.1f} %")
