#!/bin/bash
# bench under a few scheduling knobs (C2, three / four ensembles in flight)
OUT=gpurun_out; mkdir -p $OUT; L=$OUT/knobs.log; rm -f $L
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras ${EXTRA} > $OUT/knob_$tag.json 2> $OUT/knob_$tag.err
  python - <<PY >> $L
import json
d=json.load(open("$OUT/knob_$tag.json"))
o=d.get("one_ensemble_at_a_time")
print("$tag", round(d["value"],1), round(d["ms_per_step"],2), o and round(o["value"],1), {k:round(x,1) for k,x in d["stage_ms"].items() if x>5})
PY
}
EXTRA="" run base A=1
EXTRA="" run noteam KBDM_TEAM_HQR=0
EXTRA="" run team300 KBDM_TEAM_MIN_L=300
EXTRA="--in-flight 4" run base4 A=1
EXTRA="--in-flight 4" run noteam4 KBDM_TEAM_HQR=0
EXTRA="" run frac40 KBDM_LANE0_FRAC=0.4
EXTRA="" run frac60 KBDM_LANE0_FRAC=0.6
cat $L
