"""Condense the rocprofv3 outputs of tools/collect_profiles.sh into the two small files kept under
profiles/: <tag>_kernel_stats.csv (per-kernel calls / total / average / min / max) and
<tag>_pmc_traffic.json (HBM bytes per launch from FETCH_SIZE / WRITE_SIZE, KiB units, FETCH x2
gfx950 correction as MI355X_MICROARCH.md prescribes)."""
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def short(name):
    m = re.match(r"(void )?([\w:<>]+?)(<[^(]*>)?\(", name)
    base = name.split("(")[0].replace("void ", "")
    return base


def find(sub, pattern):
    files = glob.glob(os.path.join(out, f"{tag}_{sub}", "**", pattern), recursive=True)
    return files[0] if files else None


stats = find("trace", "*kernel_stats.csv")
if stats:
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as f:
        f.write("kernel,calls,total_ms,avg_ms,min_ms,max_ms,percent\n")
        for r in rows:
            f.write(f"{short(r['Name'])},{r['Calls']},{float(r['TotalDurationNs'])/1e6:.3f},{float(r['AverageNs'])/1e6:.4f},"
                    f"{float(r['MinNs'])/1e6:.4f},{float(r['MaxNs'])/1e6:.4f},{r['Percentage']}\n")

acc = {}
for sub, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    fn = find(sub, "*counter_collection.csv")
    if not fn:
        continue
    per = {}
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] != key:
            continue
        k = short(r["Kernel_Name"])
        d = per.setdefault(k, {})
        d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    for k, d in per.items():
        acc.setdefault(k, {})[key] = (sum(d.values()) / len(d), len(d))
kern = {}
for k, v in acc.items():
    fe, nf = v.get("FETCH_SIZE", (0.0, 0))
    wr, nw = v.get("WRITE_SIZE", (0.0, 0))
    kern[k] = {"FETCH_SIZE_KiB_per_launch": fe, "WRITE_SIZE_KiB_per_launch": wr, "launches_seen": max(nf, nw),
               "hbm_bytes_per_launch": (2.0 * fe + wr) * 1024.0}
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate runs of `python bench.py --steps 3 "
                   "--warmup 1` (C2, 151 members, two lanes). Counter unit: KiB. hbm_bytes_per_launch = (2*FETCH_SIZE + "
                   "WRITE_SIZE)*1024: the x2 is the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md (exact for 16 B/lane "
                   "streaming reads; other access widths are uncalibrated, so treat as an estimate). Kernels launched once "
                   "per lane appear with the average over both lanes' launches.",
           "kernels": kern}, open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print("wrote", tag)
