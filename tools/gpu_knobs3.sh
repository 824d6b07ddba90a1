#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; L=$OUT/knobs3.log; rm -f $L
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-extras ${EXTRA} > $OUT/knob_$tag.json 2> $OUT/knob_$tag.err
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
EXTRA="" run base A=1
EXTRA="" run lanes3 KBDM_LANES=3
EXTRA="--in-flight 2" run lanes3_if2 KBDM_LANES=3
EXTRA="" run lanes1 KBDM_LANES=1
EXTRA="--in-flight 4" run lanes1_if4 KBDM_LANES=1
EXTRA="" run teammax KBDM_TEAM_MIN_L=256
cat $L
