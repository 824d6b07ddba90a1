#!/bin/bash
# Run ON THE GPU BOX: per-kernel CU-busy cycles / wave cycles / VALU instructions for one C2 ensemble at a time
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/occ
timeout -k 10 400 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/occ -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --in-flight 1 > /dev/null 2> $OUT/occ.log
