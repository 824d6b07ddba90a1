#!/bin/bash
# round-2 first check of the second-generation QR iteration: parity tests, then single-member and bench timings old vs new
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/r2a_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/r2a_pytest.log
tail -5 $OUT/r2a_pytest.log
for v in 1 2; do
  echo "== HQR_V=$v single m=400" | tee -a $OUT/r2a_stage.log
  KBDM_HQR_V=$v KBDM_HQR_PROF=1 timeout -k 10 120 python tools/stage_one.py 400 >> $OUT/r2a_stage.log 2>&1
  KBDM_HQR_V=$v KBDM_HQR_PROF=1 KBDM_TEAM_HQR=0 timeout -k 10 120 python tools/stage_one.py 400 >> $OUT/r2a_stage.log 2>&1
  KBDM_HQR_V=$v timeout -k 10 120 python tools/stage_one.py 400 300 200 100 >> $OUT/r2a_stage.log 2>&1
done
cat $OUT/r2a_stage.log
for v in 1 2; do
  KBDM_HQR_V=$v timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/r2a_bench_v$v.json 2> $OUT/r2a_bench_v$v.err; echo "bench v$v rc=$?"
  python - <<PY
import json
d=json.load(open("$OUT/r2a_bench_v$v.json"))
print("v$v", d["value"], d["ms_per_step"], d.get("one_ensemble_at_a_time"), {k:round(x,1) for k,x in d["stage_ms"].items()})
PY
done
