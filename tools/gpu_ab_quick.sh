#!/bin/bash
# bench with a knob off / on, no test suite:  tools/gpu_ab_quick.sh TAG KNOB [v0 v1]
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-abq}; KNOB=${2:-KBDM_BIDIAG_FUSED}; V0=${3:-0}; V1=${4:-1}
for v in $V0 $V1 $V0 $V1; do
  env $KNOB=$v timeout -k 10 200 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras > $OUT/${TAG}_b$v.json 2> $OUT/${TAG}_b$v.err
  python - <<PY
import json
d=json.load(open("$OUT/${TAG}_b$v.json"))
o=d.get("one_ensemble_at_a_time")
print("$KNOB=$v", round(d["value"],1), round(d["ms_per_step"],2), o and round(o["value"],1), {k:round(x,1) for k,x in d["stage_ms"].items() if x>2})
PY
done
