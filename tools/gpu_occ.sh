#!/bin/bash
# per-kernel VALU instructions / wave cycles / CU-busy (one ensemble at a time) + summary
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/occ
timeout -k 10 400 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/occ -- python3 $ROOT/bench.py --steps 2 --warmup 1 --trace-mode --in-flight 1 > /dev/null 2> $OUT/occ.log
rm -rf $OUT/occ2
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/occ2 -- python3 $ROOT/bench.py --steps 2 --warmup 1 --trace-mode --in-flight 1 > /dev/null 2> $OUT/occ2.log
cd $ROOT
python3 - <<'PY' | tee $OUT/occ_summary.txt
import csv, glob, collections
for d in ("occ", "occ2"):
    fns = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); seen = set()
    for fn in fns:
        for r in csv.DictReader(open(fn)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
            k = (fn, r["Dispatch_Id"])
            if k not in seen: seen.add(k); calls[name] += 1
    ctrs = sorted({c for v in acc.values() for c in v})
    nens = max(1, calls.get("k_dc_sv", 2) // 2)
    print(d, "ensembles:", nens, "counters per ensemble (G):", ctrs)
    key = "SQ_INSTS_VALU"
    for name, v in sorted(acc.items(), key=lambda kv: -kv[1][key])[:22]:
        print(f"{name:22s} {calls[name]//nens:5d} " + " ".join(f"{v[c]/nens/1e9:9.3f}" for c in ctrs))
    print("TOTAL".ljust(28) + " ".join(f"{sum(v[c] for v in acc.values())/nens/1e9:9.3f}" for c in ctrs))
PY
