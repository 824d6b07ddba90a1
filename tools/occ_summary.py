"""Per-kernel CU-busy time from tools/pmc_occupancy.sh (rocprofv3 --pmc SQ_BUSY_CU_CYCLES ...): sum over launches,
in ms of the whole chip (busy CU-cycles / 256 CUs / clock)."""
import csv, glob, os, re, sys, collections
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "occ")
fn = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(fn)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); calls[name] += 1
tot = sum(v["SQ_BUSY_CU_CYCLES"] for v in acc.values())
print("kernel,launches,busy_cu_share,valu_insts_G,wave_cycles_G")
for name, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_BUSY_CU_CYCLES"]):
    print(f"{name},{calls[name]},{v['SQ_BUSY_CU_CYCLES']/tot:.3f},{v['SQ_INSTS_VALU']/1e9:.2f},{v['SQ_WAVE_CYCLES']/1e9:.2f}")
