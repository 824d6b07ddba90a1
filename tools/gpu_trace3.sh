#!/bin/bash
# kernel timeline of the default bench (three ensembles in flight)
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/trace3
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace3 -- python3 $ROOT/bench.py --steps 9 --warmup 3 --no-cpu-baseline --no-extras > $OUT/trace3.json 2> $OUT/trace3.log
ls -la $OUT/trace3/*/
