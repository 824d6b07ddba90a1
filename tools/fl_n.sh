#!/bin/bash
# Run ON THE GPU BOX: the in-flight headline for several numbers of ensembles in flight.  usage: tools/fl_n.sh "ENV=.." n n n
ENVV=$1; shift
for n in "$@"; do
  val=$(env $ENVV timeout -k 10 200 python bench.py --steps 48 --warmup 6 --trace-mode --in-flight $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s  %.2f ms/step' % (d['value'], d['ms_per_step']))")
  echo "[$ENVV in flight $n] $val"
done
