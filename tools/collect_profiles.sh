#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:
#   bash tools/collect_profiles.sh <tag>
# Four separate rocprofv3 passes of the same bench command (kernel trace + stats; PMC FETCH_SIZE;
# PMC WRITE_SIZE; PMC MFMA busy cycles - counters are collected on their own, MI355X_MICROARCH.md), then the
# summaries are written to gpurun_out/<tag>_* ; copy them into profiles/ afterwards.
set -e
TAG=${1:-r1}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 12 --warmup 4 --trace-mode"
rm -rf $OUT/${TAG}_trace $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_mfma
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- $CMD > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_trace.log
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_fetch -- $CMD > /dev/null 2> $OUT/${TAG}_fetch.log
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_write -- $CMD > /dev/null 2> $OUT/${TAG}_write.log
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${TAG}_mfma -- $CMD --in-flight 1 > /dev/null 2> $OUT/${TAG}_mfma.log
cd $ROOT
python3 tools/summarise_profiles.py $TAG
# (every pass is all-in-flight-4 or - the MFMA pass - all-in-flight-1: bench.py --trace-mode)
# (the MFMA pass runs one ensemble at a time so that a launch's busy cycles are not diluted by the other ensemble)
python3 tools/mfma_util.py $(find $OUT/${TAG}_mfma -name "*kernel_trace.csv" | head -1) $(find $OUT/${TAG}_mfma -name "*counter_collection.csv" | head -1) > $OUT/${TAG}_north_star_kernels.json
