#!/bin/bash
# Run ON THE GPU BOX: MFMA utilisation of the rank-64 updates on a workload whose launches fill the chip (C3: 1024 members of
# m = 512), PMC pass on its own (one ensemble at a time).  bash tools/mfma_c3.sh <tag>
TAG=${1:-r4}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/${TAG}_mfma_c3
timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${TAG}_mfma_c3 -- python3 $ROOT/bench.py --workload C3 --steps 1 --warmup 1 --trace-mode --in-flight 1 > /dev/null 2> $OUT/${TAG}_mfma_c3.log
cd $ROOT
python3 tools/mfma_util.py $(find $OUT/${TAG}_mfma_c3 -name "*kernel_trace.csv" | head -1) $(find $OUT/${TAG}_mfma_c3 -name "*counter_collection.csv" | head -1) C3 > $OUT/${TAG}_north_star_kernels_c3.json
cat $OUT/${TAG}_north_star_kernels_c3.json
