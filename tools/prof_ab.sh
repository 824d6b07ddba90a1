#!/bin/bash
# GPU box: per-launch durations of the Aberth kernels for one workload (default C1: three noise-free members)
W=${1:-C1}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/prof_ab
KBDM_EIG_AB=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_ab -- python3 $ROOT/tools/check_ab.py $W > $OUT/prof_ab.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv, glob
fn = glob.glob("gpurun_out/prof_ab/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(fn))]
ab = [r for r in rows if r["Kernel_Name"].startswith("k_ab_iter")]
# the last execute of the first engine: take the last 3 * 24 * levels launches before the first k_hqr2 of ... simply print per-launch durations of the last block of consecutive k_ab_iter launches
runs, cur = [], []
for r in rows:
    if r["Kernel_Name"].startswith("k_ab_iter"): cur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    elif cur: runs.append(cur); cur = []
if cur: runs.append(cur)
runs = [x for x in runs if len(x) >= 24]
last = runs[2] if len(runs) > 2 else runs[-1]
for lv in range(0, len(last), 24):
    print("level", lv // 24, "us per launch:", [round(x) for x in last[lv:lv + 24]])
print("sum ms:", sum(last) / 1e3)
for name in ("k_ab_leaf", "k_ab_finish", "k_hqr2"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if r["Kernel_Name"].startswith(name)]
    if d: print(name, "avg us", sum(d) / len(d), "n", len(d))
PY
