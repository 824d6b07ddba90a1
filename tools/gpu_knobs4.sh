#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; L=$OUT/knobs4.log; rm -f $L
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras ${EXTRA} > $OUT/knob_$tag.json 2> $OUT/knob_$tag.err
  python - <<PY >> $L
import json
try:
    d=json.load(open("$OUT/knob_$tag.json"))
    o=d.get("one_ensemble_at_a_time")
    print("$tag", round(d["value"],1), round(d["ms_per_step"],2), o and round(o["value"],1), {k:round(x,1) for k,x in d["stage_ms"].items() if x>5})
except Exception as e:
    print("$tag failed", e)
PY
}
for rep in 1 2; do
EXTRA="" run t192_$rep KBDM_TEAM_MIN_L=192
EXTRA="" run t256_$rep KBDM_TEAM_MIN_L=256
EXTRA="" run t300_$rep KBDM_TEAM_MIN_L=300
EXTRA="" run t360_$rep KBDM_TEAM_MIN_L=360
done
cat $L
