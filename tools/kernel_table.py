"""profiles/<tag>_kernel_table.md from <tag>_kernel_stats.csv and <tag>_pmc_traffic.json (both written by
tools/collect_profiles.sh + tools/summarise_profiles.py): time share, average launch and HBM bytes per launch of the
largest kernels of the traced bench run.  `python tools/kernel_table.py <tag> [dir]`"""
import csv
import json
import os
import sys

tag = sys.argv[1]
d = sys.argv[2] if len(sys.argv) > 2 else "profiles"
rows = list(csv.DictReader(open(os.path.join(d, f"{tag}_kernel_stats.csv"))))
pmc = json.load(open(os.path.join(d, f"{tag}_pmc_traffic.json")))["kernels"]
tot = sum(float(r["total_ms"]) for r in rows)
out = [f"# {tag}: kernels of the traced bench run (`bench.py --steps 12 --warmup 4 --trace-mode`: every ensemble in flight, four at a time)",
       "",
       "HBM bytes per launch from the separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes ((2 FETCH + WRITE) KiB, gfx950 correction); "
       "the rate is those bytes over the kernel's average duration in the traced run (kernels of four ensembles overlap, so it is a lower "
       "bound of what the kernel reaches alone).",
       "",
       "| kernel | launches | total ms | share | avg ms | HBM MB / launch | GB/s |", "|---|---|---|---|---|---|---|"]
for r in rows[:26]:
    k = r["kernel"]
    b = pmc.get(k, {}).get("hbm_bytes_per_launch")
    mb = "" if b is None else f"{b / 1e6:.1f}"
    gbs = "" if b is None else f"{b / (float(r['avg_ms']) * 1e-3) / 1e9:.0f}"
    out.append(f"| `{k}` | {r['calls']} | {float(r['total_ms']):.1f} | {100 * float(r['total_ms']) / tot:.1f} % | {float(r['avg_ms']):.3f} | {mb} | {gbs} |")
open(os.path.join(d, f"{tag}_kernel_table.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
