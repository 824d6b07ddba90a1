#!/bin/bash
# stage times by batch shape with a knob off / on
OUT=gpurun_out; mkdir -p $OUT; KNOB=${1:-KBDM_BIDIAG_FUSED}; L=$OUT/stage_ab.log; rm -f $L
for v in 0 1; do echo "== $KNOB=$v" >> $L; env $KNOB=$v timeout -k 10 200 python tools/stage_by_m.py >> $L 2>&1; done
cat $L
