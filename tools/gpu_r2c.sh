#!/bin/bash
# after the start-of-sweep wait of the team protocol: the failing member, all of C4, the GPU suite, the bench
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r2c}
timeout -k 10 120 python tools/check_member.py C4 1123 > $OUT/${TAG}_member.log 2>&1; cat $OUT/${TAG}_member.log
timeout -k 10 300 python tools/check_c4_status.py > $OUT/${TAG}_c4.log 2>&1; tail -5 $OUT/${TAG}_c4.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/${TAG}_pytest.log
tail -14 $OUT/${TAG}_pytest.log
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"
cat $OUT/${TAG}_bench.json
