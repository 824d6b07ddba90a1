#!/bin/bash
# C4 after a change: status words + three members against the oracle, the full-size tests, one timed C4 step
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-c4}
timeout -k 10 400 python tools/check_c4_status.py > $OUT/${TAG}_status.log 2>&1; tail -4 $OUT/${TAG}_status.log
timeout -k 10 300 python tools/check_member.py C4 1200 >> $OUT/${TAG}_status.log 2>&1; tail -3 $OUT/${TAG}_status.log
timeout -k 10 400 python bench.py --workload C4 --steps 2 --warmup 1 --in-flight 1 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python - <<PY
import json
d=json.load(open("$OUT/${TAG}_bench.json"))
print("C4", round(d["value"],1), "solves/s", round(d["ms_per_step"],1), "ms", {k:round(x) for k,x in d["stage_ms"].items() if x>20})
PY
