#!/bin/bash
# stage times by batch shape under env settings given as arguments: tools/gpu_stage_env.sh "A=1" "A=2 B=3" ...
OUT=gpurun_out; mkdir -p $OUT; L=$OUT/stage_env.log; rm -f $L
for e in "$@"; do echo "== $e" >> $L; env $e timeout -k 10 200 python tools/stage_by_m.py 2>&1 | grep -E "n 1 status|n 151" | tail -3 >> $L; done
cat $L
