#!/bin/bash
# Run ON THE GPU BOX: the in-flight headline (bench.py --trace-mode: nothing but ensembles in flight) under a few knob settings.
# usage: tools/fl_ab.sh "VAR=val VAR=val" "VAR=val" ...   (one quoted group per run; "" = defaults)
OUT=gpurun_out; mkdir -p $OUT
for grp in "$@"; do
  val=$(env $grp timeout -k 10 200 python bench.py --steps 40 --warmup 4 --trace-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s  %.2f ms/step' % (d['value'], d['ms_per_step']))")
  echo "[$grp] $val"
done
