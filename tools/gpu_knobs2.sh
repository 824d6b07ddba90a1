#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; L=$OUT/knobs2.log; rm -f $L
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras ${EXTRA} > $OUT/knob_$tag.json 2> $OUT/knob_$tag.err
  python - <<PY >> $L
import json
d=json.load(open("$OUT/knob_$tag.json"))
o=d.get("one_ensemble_at_a_time")
print("$tag", round(d["value"],1), round(d["ms_per_step"],2), o and round(o["value"],1), {k:round(x,1) for k,x in d["stage_ms"].items() if x>5})
PY
}
EXTRA="" run base KBDM_BIDIAG_FUSED=0
EXTRA="" run ntfac512 KBDM_BIDIAG_FUSED=0 KBDM_NT_FAC=512
EXTRA="" run fused KBDM_BIDIAG_FUSED=1
EXTRA="" run fused_ntfac512 KBDM_BIDIAG_FUSED=1 KBDM_NT_FAC=512
EXTRA="" run invit512 KBDM_BIDIAG_FUSED=0 KBDM_INVIT_REG=0
EXTRA="--in-flight 2" run base2 KBDM_BIDIAG_FUSED=0
cat $L
