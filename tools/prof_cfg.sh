#!/bin/bash
# Run ON THE GPU BOX: kernel stats of one full-size configuration (tools/run_configs.py <name>)
NAME=${1:-C5}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/cfg_${NAME}_trace
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg_${NAME}_trace -- python3 $ROOT/tools/run_configs.py $NAME > $OUT/cfg_${NAME}.json 2> $OUT/cfg_${NAME}.log
cat $OUT/cfg_${NAME}.json
head -14 $(find $OUT/cfg_${NAME}_trace -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4,8 | cut -c1-120
