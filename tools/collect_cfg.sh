#!/bin/bash
# Run ON THE GPU BOX: kernel trace + stats of ONE other BASELINE configuration (C3 / C4 / C5 / NS), one ensemble at a time:
#   bash tools/collect_cfg.sh <tag> <workload>      -> gpurun_out/<tag>_<workload>_kernel_stats.csv (copy into profiles/)
TAG=${1:-r4}; WL=${2:-C4}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
D=$OUT/${TAG}_${WL}_trace; rm -rf $D
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $ROOT/bench.py --workload $WL --steps 1 --warmup 1 --trace-mode --in-flight 1 > $OUT/${TAG}_${WL}_bench_under_rocprof.json 2> $OUT/${TAG}_${WL}_trace.log
cd $ROOT
python3 - "$D" "$OUT/${TAG}_${WL}_kernel_stats.csv" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kb::", "")
    d[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6; n[k] += 1
tot = sum(d.values())
with open(sys.argv[2], "w") as o:
    o.write("kernel,calls,total_ms,avg_ms,percent\n")
    for k, v in sorted(d.items(), key=lambda kv: -kv[1]):
        o.write("%s,%d,%.3f,%.4f,%.2f\n" % (k, n[k], v, v / n[k], 100 * v / tot))
print(open(sys.argv[2]).read()[:1500])
PY
