#!/bin/bash
# Run ON THE GPU BOX: kernel trace + stats of four C2 ensembles one at a time; prints the top kernels.  tag, then env assignments
TAG=${1:-t}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rm -rf $OUT/${TAG}_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 $ROOT/tools/lone_once.py 4 > $OUT/${TAG}_trace.log 2>&1
f=$(find $OUT/${TAG}_trace -name '*kernel_stats.csv' | head -1)
echo "== $TAG ($*)"; head -14 "$f" | cut -d, -f1-5 | sed 's/(.*)//'
