#!/bin/bash
# Run ON THE GPU BOX: one PMC pass (MFMA busy cycles) of the bench command -> gpurun_out/<tag>_mfma
TAG=${1:-r1}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/${TAG}_mfma
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${TAG}_mfma -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --in-flight 1 > /dev/null 2> $OUT/${TAG}_mfma.log
ls $OUT/${TAG}_mfma/*/ | head
