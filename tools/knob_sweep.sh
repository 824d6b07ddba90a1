#!/bin/bash
# GPU box: bench (12 steps) under a few settings of one environment knob:  tools/knob_sweep.sh KNOB v1 v2 ...
KNOB=$1; shift
for v in "$@"; do
  env $KNOB=$v timeout -k 10 200 python bench.py --steps 24 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/ks.json 2> gpurun_out/ks.err
  python - <<PY
import json
d=json.load(open("gpurun_out/ks.json")); o=d.get("one_ensemble_at_a_time")
print("$KNOB=$v", round(d["value"],1), round(d["ms_per_step"],2), o and round(o["value"],1))
PY
done
