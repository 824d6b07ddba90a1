"""The two figures BASELINE.json's north_star asks for besides the headline, from the rocprofv3 outputs of
tools/collect_profiles.sh (workload C2):
  * MFMA utilisation of the rank-64 trailing updates of the two blocked reductions (k_trail_update = the SVD panel
    update, k_hess_update): (a) SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs) from the
    PMC pass, where every kernel runs on its own; (b) algorithmic flops / summed launch durations from the kernel
    trace of the normal (overlapped) run, against the 78.6 TFLOP/s FP64 matrix peak;
  * HBM rate of the Hankel build k_hankel: algorithmic bytes (16 m^2 written + 32 m read per member: inside the
    pipeline only U^{p-1} is materialised) / launch duration, against 8 TB/s.
usage: python tools/mfma_util.py <kernel_trace.csv> [<counter_collection.csv> [C3]]   (C3: the trace is of bench.py --workload C3)"""
import csv
import json
import sys

PEAK_TFLOPS = 78.6
NB, NX = 32, 64
MS = list(range(100, 401, 2))
if len(sys.argv) > 3 and sys.argv[3] == "C3":
    MS = [512] * 1024


def npanels(m):
    return (m - NX) // NB if m >= NB + NX else 0


def flops_svd(m):       # C[nn x nn] -= [V|X] [Y|U]^H, K = 64, per panel
    return sum(8.0 * 2 * NB * (m - (p + 1) * NB) ** 2 for p in range(npanels(m)))


def flops_hess(m):      # W[:, cbase:] -= [Y|V] [V|Z]^H, all n rows
    return sum(8.0 * 2 * NB * m * (m - (p + 1) * NB) for p in range(npanels(m)))


rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = len([r for r in rows if r["Kernel_Name"].startswith("k_hankel")]) / 2.0     # two lanes per step
out = {"steps_in_trace": nsteps}
for kname, fl in (("k_trail_update", sum(flops_svd(m) for m in MS)), ("k_hess_update", sum(flops_hess(m) for m in MS))):
    sel = [r for r in rows if r["Kernel_Name"].startswith(kname)]
    t = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel) * 1e-9
    out[kname] = {"launches": len(sel), "algorithmic_gflop_per_step": fl / 1e9, "summed_ms_per_step": 1e3 * t / nsteps,
                  "tflops_over_launch_time": fl * nsteps / t / 1e12, "frac_of_fp64_matrix_peak": fl * nsteps / t / 1e12 / PEAK_TFLOPS}
hk = [r for r in rows if r["Kernel_Name"].startswith("k_hankel")]
t = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in hk) * 1e-9
b = sum(16.0 * m * m + 32.0 * m for m in MS)
out["k_hankel"] = {"launches": len(hk), "algorithmic_MB_per_step": b / 1e6, "GBps": b * nsteps / t / 1e9,
                   "frac_of_hbm_peak": b * nsteps / t / 8e12}
if len(sys.argv) > 2:
    acc = {}
    for r in csv.DictReader(open(sys.argv[2])):
        k = r["Kernel_Name"].split("(")[0]
        acc.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    # per launch (the PMC pass runs every kernel on its own): the big first-panel launches against the small last ones
    per = {}
    for r in csv.DictReader(open(sys.argv[2])):
        k = r["Kernel_Name"].split("(")[0]
        if k in ("k_trail_update", "k_hess_update"):
            d = per.setdefault((k, r["Dispatch_Id"]), {"wgs": int(r["Grid_Size"]) // 256})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for k in ("k_trail_update", "k_hess_update"):
        ls = sorted((d for (kk, _), d in per.items() if kk == k and d.get("GRBM_GUI_ACTIVE")), key=lambda d: -d["wgs"])
        if ls:
            util = [d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4) for d in ls]
            big = [u for d, u in zip(ls, util) if d["wgs"] == ls[0]["wgs"]]
            small = [u for d, u in zip(ls, util) if d["wgs"] == ls[-1]["wgs"]]
            out[k]["pmc_mfma_utilisation_largest_launches"] = {"workgroups": ls[0]["wgs"], "mean": sum(big) / len(big), "max": max(big)}
            out[k]["pmc_mfma_utilisation_smallest_launches"] = {"workgroups": ls[-1]["wgs"], "mean": sum(small) / len(small)}
    for k in ("k_trail_update", "k_hess_update"):
        v = acc.get(k)
        if v and v.get("GRBM_GUI_ACTIVE"):
            out[k]["pmc_mfma_busy_cycles"] = v["SQ_VALU_MFMA_BUSY_CYCLES"]
            out[k]["pmc_mfma_utilisation"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4)
print(json.dumps(out, indent=1))
