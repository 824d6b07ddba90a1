#!/bin/bash
# The default bench under environment / option settings, one line per setting (run ON THE GPU BOX through gpurun):
#   bash tools/gpu_knobs.sh "A=1" "KBDM_TEAM_HQR=0" "KBDM_LANES=3 EXTRA=--in-flight=2" ...
# (EXTRA=... is passed to bench.py as an option, everything else is exported)
OUT=gpurun_out; mkdir -p $OUT; L=$OUT/knobs.log; rm -f $L
n=0
for e in "$@"; do
  n=$((n + 1)); extra=""; envs=""
  for w in $e; do case $w in EXTRA=*) extra="$extra ${w#EXTRA=}";; *) envs="$envs $w";; esac; done
  env $envs timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras $extra > $OUT/knob_$n.json 2> $OUT/knob_$n.err
  python - <<PY >> $L
import json
try:
    d = json.load(open("$OUT/knob_$n.json"))
    o = d.get("one_ensemble_at_a_time")
    print("$e |", round(d["value"], 1), "solves/s", round(d["ms_per_step"], 2), "ms/step, one at a time", o and round(o["value"], 1),
          {k: round(x, 1) for k, x in d["stage_ms"].items() if x > 5})
except Exception as ex:
    print("$e | failed", ex)
PY
done
cat $L
