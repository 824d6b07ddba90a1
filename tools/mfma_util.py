"""MFMA utilisation of the rank-64 trailing updates and HBM rate of the Hankel build from a rocprofv3 kernel trace
of `bench.py` (C2).  usage: python tools/mfma_util.py <kernel_trace.csv> [lane0_members]"""
import csv, json, sys
import numpy as np

PEAK = 78.6e12
NB = 32
ms_all = np.arange(100, 401, 2)[::-1]            # sorted by size, largest first (plan order)
n0 = int(sys.argv[2]) if len(sys.argv) > 2 else 32
lanes = {32 if n0 == 32 else n0: ms_all[:n0], len(ms_all) - n0: ms_all[n0:]}

def npanels(m):
    return max(0, (m - 64) // NB) if m >= 2 * 64 else max(0, (m - 64) // NB)

rows = list(csv.DictReader(open(sys.argv[1])))
out = {}
for kname, kind in (("k_trail_update", "svd"), ("k_hess_update", "hess")):
    sel = [r for r in rows if r["Kernel_Name"].startswith(kname)]
    # the panel index of a launch: launches of one lane come in panel order; identify the lane by Grid_Size_Z
    per = {}
    tot_f = tot_t = 0.0
    seen = {}
    for r in sel:
        z = int(r["Grid_Size_Z"])
        members = lanes.get(z)
        if members is None:
            continue
        gx = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
        gy = int(r["Grid_Size_Y"])
        # panel from the grid: trailing size of the largest member
        mmax = int(members[0])
        if kind == "svd":
            # grid = ceil((mmax - p0 - NB) / 64)
            cands = [p for p in range(0, 40) if (mmax - p * NB - NB + 63) // 64 == gx and mmax - p * NB - NB > 0]
        else:
            cands = [p for p in range(0, 40) if (mmax - (p + 1) * NB + 63) // 64 == gy and mmax - (p + 1) * NB > 0]
        key = (z, gx, gy)
        idx = seen.get(key, 0)
        seen[key] = idx + 1
        nsteps = len([1 for r2 in sel if int(r2["Grid_Size_Z"]) == z and int(r2["Grid_Size_X"]) // int(r2["Workgroup_Size_X"]) == gx and int(r2["Grid_Size_Y"]) == gy])
        per_step = max(1, len(cands))
        p = cands[(idx % per_step)] if cands else 0
        fl = 0.0
        for m in members:
            m = int(m)
            if kind == "svd":
                nn = m - p * NB - NB
                if nn > 0 and p < (m - 64) // NB + (1 if (m - 64) % NB == 0 and False else 0) + 1:
                    fl += 8.0 * 2 * NB * nn * nn
            else:
                nc = m - (p + 1) * NB
                if nc > 0:
                    fl += 8.0 * 2 * NB * m * nc
        dt = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
        tot_f += fl
        tot_t += dt
    out[kname] = {"launches": len(sel), "tflops": tot_f / tot_t / 1e12 if tot_t else None,
                  "frac_of_fp64_mfma_peak": tot_f / tot_t / PEAK if tot_t else None, "total_ms": 1e3 * tot_t}
hk = [r for r in rows if r["Kernel_Name"].startswith("k_hankel")]
b = t = 0.0
for r in hk:
    members = lanes.get(int(r["Grid_Size_Z"]))
    if members is None:
        continue
    b += sum(16.0 * int(m) * int(m) + 16.0 * (2 * int(m)) for m in members)
    t += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
out["k_hankel"] = {"launches": len(hk), "GBps": b / t / 1e9 if t else None, "frac_of_hbm_peak": b / t / 8e12 if t else None}
print(json.dumps(out, indent=1))
