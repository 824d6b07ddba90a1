#!/bin/bash
# Run ON THE GPU BOX: HBM read traffic (FETCH_SIZE) per kernel of one full-size configuration
NAME=${1:-C5}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/cfg_${NAME}_fetch
timeout -k 10 800 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/cfg_${NAME}_fetch -- python3 $ROOT/tools/run_configs.py $NAME > $OUT/cfg_${NAME}_fetch.json 2> $OUT/cfg_${NAME}_fetch.log
cat $OUT/cfg_${NAME}_fetch.json
