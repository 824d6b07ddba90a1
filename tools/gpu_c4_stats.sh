#!/bin/bash
# per-kernel totals of one C4 step (1001 members, N = 4096, m = 200..1200)
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/c4trace
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4trace -- python3 $ROOT/bench.py --workload C4 --steps 1 --warmup 1 --in-flight 1 --no-cpu-baseline --no-extras > $OUT/c4trace.json 2> $OUT/c4trace.log
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
fn = glob.glob("gpurun_out/c4trace/**/*kernel_stats.csv", recursive=True)
print(fn)
rows = list(csv.DictReader(open(fn[0])))
for r in rows[:24]:
    print(r["Name"].split("(")[0][:28].ljust(30), r["Calls"].rjust(6), "%10.1f ms" % (float(r["TotalDurationNs"]) / 1e6), r["Percentage"])
PY
