#!/bin/bash
# Run ON THE GPU BOX: kernel trace + stats of one bench command into gpurun_out/<tag>_trace
TAG=${1:-t}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/${TAG}_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_trace.log
