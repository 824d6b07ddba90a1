#!/bin/bash
# Run ON THE GPU BOX: the Hankel build alone under rocprofv3 (tools/hankel_bw.py); prints GB/s per kernel variant
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/hankel_trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/hankel_trace -- python3 $ROOT/tools/hankel_bw.py > $OUT/hankel_trace.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/hankel_trace/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "hankel" in r["Kernel_Name"]:
        d[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
byt = 384 * (16.0 * 512 * 512 + 16 * 1023)
for k, v in d.items():
    v.sort()
    print("%-28s n=%d  min %.3f ms  median %.3f ms  -> %.0f GB/s (median)  %.0f GB/s (best)" % (k, len(v), v[0], v[len(v) // 2], byt / v[len(v) // 2] / 1e6, byt / v[0] / 1e6))
PY
tail -2 $OUT/hankel_trace.log
