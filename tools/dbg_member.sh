#!/bin/bash
# one member of a config under several settings of the QR iteration (debug)
OUT=gpurun_out; mkdir -p $OUT; CFG=${1:-C4}; M=${2:-1123}
L=$OUT/dbg_member_${CFG}_${M}.log; rm -f $L
run() { echo "== $*" >> $L; env "$@" timeout -k 10 120 python tools/check_member.py $CFG $M >> $L 2>&1; }
run
run KBDM_TEAM_HQR=0
run KBDM_NB_HQR2=6
run KBDM_NB_HQR2=4
run KBDM_NB_HQR2=7
run KBDM_TEAM_HQR=0 KBDM_NB_HQR2=6
run KBDM_HQR_PROF=1
cat $L
