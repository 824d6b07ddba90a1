#!/bin/bash
# second-generation QR iteration: parity tests, single-member profile (team / solo), bench
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r2b}
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/${TAG}_pytest.log
tail -3 $OUT/${TAG}_pytest.log
rm -f $OUT/${TAG}_stage.log
KBDM_HQR_PROF=1 timeout -k 10 120 python tools/stage_one.py 400 >> $OUT/${TAG}_stage.log 2>&1
KBDM_HQR_PROF=1 KBDM_TEAM_HQR=0 timeout -k 10 120 python tools/stage_one.py 400 >> $OUT/${TAG}_stage.log 2>&1
timeout -k 10 120 python tools/stage_one.py 400 300 200 100 >> $OUT/${TAG}_stage.log 2>&1
cat $OUT/${TAG}_stage.log
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$OUT/${TAG}_bench.json"))
print("bench", d["value"], d["ms_per_step"], d.get("one_ensemble_at_a_time"), {k:round(x,1) for k,x in d["stage_ms"].items()})
PY
