#!/bin/bash
# per-launch durations of the panel kernels for one m = 400 member, fused off / on
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
rm -rf $OUT/ptrace$v
KBDM_BIDIAG_FUSED=$v timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/ptrace$v -- python3 $ROOT/tools/stage_one.py 400 > $OUT/ptrace$v.log 2>&1
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
for v in (0, 1):
    fn = glob.glob(f"gpurun_out/ptrace{v}/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(fn)))
    per = collections.defaultdict(list)
    for r in rows:
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        per[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("FUSED", v)
    for n in ("k_bidiag_panel<0>", "k_bidiag_panel<1>", "k_trail_update", "k_svd_fac", "k_hess_panel", "k_hess_z", "k_hess_update", "k_hess"):
        if n in per:
            d = per[n]; k = len(d) // 2            # two executes: take the second
            print(f"  {n:20s}", " ".join(f"{x:7.0f}" for x in d[k:]), "us  sum %.0f" % sum(d[k:]))
PY
