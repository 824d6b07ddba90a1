#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/sweep_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sweep_trace -- python3 $ROOT/tools/prof_sweep.py > $OUT/sweep_trace.log 2>&1
grep -v "^W2026\|^E2026" $OUT/sweep_trace.log | tail -8
f=$(find $OUT/sweep_trace -name '*kernel_stats.csv' | head -1)
grep "k_knn\|k_prim\|k_silh" "$f" | cut -d, -f1-6
