"""GPU: how many root tiles (64 roots each) are still iterating in every launch of the Ehrlich-Aberth eigenvalue path
(k_ab_iter), per step of the tree and iteration, for C2, a C3 sample and a C4 sample -> profiles/<tag>_ab_iterations.json.
The evidence behind KB_AB_INNER_BUDGET = 4 (levels below the root stop after four iterations) and KB_AB_BUDGET = 24: the root
level's tail, the fallbacks (members handed to the QR iteration) and the eigenvalue agreement with the QR iteration.
`python tools/ab_hist.py [tag]`"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from llckbdm_amd import datasets                     # noqa: E402
from llckbdm_amd.engine import Engine                # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r4"
    works = {"C2": datasets.config2(seed=0),
             "C3 (128 of 1024 draws)": datasets.config3(count=128, m=512),
             "C4 (every 8th member)": tuple(x[::8] if i else x for i, x in enumerate(datasets.config4())),
             "C5 (4 of 64 voxels)": datasets.config5(voxels=4)}
    out = {}
    eng = Engine(0, in_flight=1)
    for name, (sigs, sidx, ms) in works.items():
        plan = eng.plan(sigs.shape[0], sigs.shape[1], sidx, ms, ms, dwell=datasets.DWELL)
        plan.upload(sigs)
        plan.execute()
        plan.ab_stats()
        plan.execute()
        st = plan.ab_stats()
        res = plan.download()
        rows = [r.tolist() for r in st if r.any()]
        nroot = rows[-1] if rows else []
        out[name] = {"members": int(len(ms)), "m_min": int(ms.min()), "m_max": int(ms.max()),
                     "fallbacks_to_qr": plan.eig_fallbacks(), "status_nonzero": int((res.status != 0).sum()),
                     "tiles_iterating_per_step_and_iteration": rows,
                     "note": "one row per step of the tree (leaf side first, the ROOT level last); column = iteration (launch)",
                     "root_level_last_busy_iteration": int(max([i for i, v in enumerate(nroot) if v] or [0])) + 1}
        print(name, "fallbacks", out[name]["fallbacks_to_qr"], "root level busy until iteration", out[name]["root_level_last_busy_iteration"])
        for r in rows:
            print("   ", r)
        plan.close()
    eng.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"{tag}_ab_iterations.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
