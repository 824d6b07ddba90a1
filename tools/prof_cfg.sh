#!/bin/bash
# Run ON THE GPU BOX: kernel-time summary of one workload (C3 / C4 / C5 / C2): bash tools/prof_cfg.sh C3 [steps]
W=${1:-C3}; STEPS=${2:-2}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/prof_$W
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$W -- python3 $ROOT/bench.py --workload $W --steps $STEPS --warmup 1 --no-cpu-baseline --no-extras > $OUT/prof_${W}_bench.json 2> $OUT/prof_$W.log
cd $ROOT
python3 - "$W" <<'PY'
import csv, glob, sys, collections
w = sys.argv[1]
fn = glob.glob(f"gpurun_out/prof_{w}/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(fn)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{w}: total kernel time {tot/1e6:.1f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    print(f'{r["Name"].split("(")[0][:40]:40s} calls {int(r["Calls"]):6d} total {float(r["TotalDurationNs"])/1e6:9.2f} ms  avg {float(r["AverageNs"])/1e3:10.1f} us  {100*float(r["TotalDurationNs"])/tot:5.1f} %')
PY
