#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
# KBDM_AB_DBG: 2 no triangle, 4 no panel row stores (results are wrong then - timing only); tools/ab_phases.py (8) gives the split directly
for dbg in 0 2 4 6; do
rm -rf $OUT/prof_ab
KBDM_AB_DBG=$dbg KBDM_EIG_AB=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_ab -- python3 $ROOT/tools/check_ab.py C1 > $OUT/prof_ab.log 2>&1
python3 - $dbg <<'PY'
import csv, glob, sys
fn = glob.glob("/root/repo/gpurun_out/prof_ab/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(fn))]
d = sorted(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if r["Kernel_Name"].startswith("k_ab_iter")), reverse=True)
print("dbg", sys.argv[1], "largest k_ab_iter launches (us):", [round(x) for x in d[:6]], "| leaf:", [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if r["Kernel_Name"].startswith("k_ab_leaf")][:2])
PY
done
