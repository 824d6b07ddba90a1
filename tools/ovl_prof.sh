set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/ovl1 $OUT/ovl2
export PROBE_STEPS=8
PROBE_INFLIGHT=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ovl1 -- python3 $ROOT/tools/overlap_probe.py > $OUT/ovl1.json 2> $OUT/ovl1.log
PROBE_INFLIGHT=2 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ovl2 -- python3 $ROOT/tools/overlap_probe.py > $OUT/ovl2.json 2> $OUT/ovl2.log
cat $OUT/ovl1.json $OUT/ovl2.json
