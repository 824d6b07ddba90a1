#!/bin/bash
# GPU suite, then the bench with a knob off / on:  tools/gpu_ab.sh TAG KNOB   (e.g. KBDM_BIDIAG_FUSED)
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-ab}; KNOB=${2:-KBDM_BIDIAG_FUSED}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/${TAG}_pytest.log
tail -5 $OUT/${TAG}_pytest.log
for v in 0 1 0 1; do
  env $KNOB=$v timeout -k 10 200 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras > $OUT/${TAG}_b$v.json 2> $OUT/${TAG}_b$v.err
  python - <<PY
import json
d=json.load(open("$OUT/${TAG}_b$v.json"))
o=d.get("one_ensemble_at_a_time")
print("$KNOB=$v", round(d["value"],1), round(d["ms_per_step"],2), o and round(o["value"],1), {k:round(x,1) for k,x in d["stage_ms"].items() if x>2})
PY
done
