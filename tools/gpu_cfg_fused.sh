#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; L=$OUT/cfg_fused.log; rm -f $L
for v in 0 1; do echo "== KBDM_BIDIAG_FUSED=$v" >> $L; KBDM_BIDIAG_FUSED=$v timeout -k 10 300 python tools/run_configs.py C5 C3 >> $L 2>&1; done
cat $L
