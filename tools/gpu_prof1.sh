#!/bin/bash
# cycle-counter breakdown of the QR iteration of one m=400 member (team and solo)
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-prof1}
rm -f $OUT/${TAG}.log
KBDM_HQR_PROF=1 timeout -k 10 120 python tools/stage_one.py 400 >> $OUT/${TAG}.log 2>&1
KBDM_HQR_PROF=1 KBDM_TEAM_HQR=0 timeout -k 10 120 python tools/stage_one.py 400 >> $OUT/${TAG}.log 2>&1
cat $OUT/${TAG}.log
