"""GPU: how far every well-posed golden case is from its tolerance (max relative error of A, T2, F and absolute phase
error of the kept lines against the vectors generated from the reference)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from helpers import canonical, keep_mask
from llckbdm_amd.engine import Engine
from llckbdm_amd.kbdm import kbdm
golden = dict(np.load("tests/golden/kbdm_golden.npz", allow_pickle=True))
eng = Engine(0)
for name in ["c1", "m300", "m150", "m100", "m101", "m102", "m180l30", "m64p2", "n3m128", "n3m256", "n6m256", "n3m512"]:
    m, l, p = (int(x) for x in golden[f"{name}__meta"]); q = float(golden[f"{name}__q"][0])
    sig = golden[str(golden[f"{name}__sig"])]
    ll, info = kbdm(sig, 5e-4, m=m, p=p, l=(None if l == m else l), q=q, engine=eng)
    kept = canonical(ll[keep_mask(ll)]); want = golden[f"{name}__kept"]
    if kept.shape != want.shape:
        print(name, "kept count differs", kept.shape, want.shape); continue
    rel = [float((np.abs(kept[:, c] - want[:, c]) / np.maximum(np.abs(want[:, c]), 1e-300)).max()) for c in range(3)]
    dph = float(np.abs(np.angle(np.exp(1j * (kept[:, 3] - want[:, 3])))).max())
    extra = ""
    if name in ("c1", "m300", "m150", "m100", "m101", "m102", "m180l30", "m64p2") and len(kept) == 16:      # noise-free: the analytic truth
        from oracle import kbdm_oracle as O           # (tools/ may use the oracle: this is a diagnostic, not the product)
        T = canonical(np.asarray(O.brain_sim_params_sorted()))
        Tm = T[[int(np.argmin(np.abs(T[:, 2] - w[2]))) for w in kept]]
        for lab, arr in (("gpu", kept), ("ref", want)):
            r3 = [float((np.abs(arr[:, c] - Tm[:, c]) / np.abs(Tm[:, c])).max()) for c in range(3)]
            ph = float(np.abs(np.angle(np.exp(1j * (arr[:, 3] - Tm[:, 3])))).max())
            extra += f" | {lab} vs truth A {r3[0]:.1e} T2 {r3[1]:.1e} F {r3[2]:.1e} ph {ph:.1e}"
    print(f"{name:8s} lines {len(kept):3d}  A {rel[0]:.2e}  T2 {rel[1]:.2e}  F {rel[2]:.2e}  phase {dph:.2e}{extra}")
