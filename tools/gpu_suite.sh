#!/bin/bash
# full -m gpu suite, then the default bench (with its extras), on the GPU box
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-suite}
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/${TAG}_pytest.log
tail -16 $OUT/${TAG}_pytest.log
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"
tail -3 $OUT/${TAG}_bench.err
cat $OUT/${TAG}_bench.json
