"""GPU: the C1-like member (m = 150, noise sigma 1e-3) with the signal multiplied by powers of ten: the kept lines must be those
of the unscaled signal with the amplitudes scaled (checked against the oracle on the scaled signal where it stays finite)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from helpers import canonical, keep_mask
from llckbdm_amd import datasets
from llckbdm_amd.engine import Engine
from llckbdm_amd.kbdm import kbdm
from oracle import kbdm_oracle as O
eng = Engine(0)
sig = O.make_noisy(O.brain_sim_signal(2048), 1e-3, 3)
base, _ = kbdm(sig, 5e-4, m=150, p=1, l=None, q=0, engine=eng)
kb = canonical(base[keep_mask(base)])
for f in (1e-30, 1e30, 1e-120, 1e120, 1e-150, 1e150):
    try:
        ll, info = kbdm(sig * f, 5e-4, m=150, p=1, l=None, q=0, engine=eng)
        k = canonical(ll[keep_mask(ll)])
        if k.shape != kb.shape:
            print(f, "kept", k.shape, "vs", kb.shape); continue
        ra = np.abs(k[:, 0] / f - kb[:, 0]).max() / np.abs(kb[:, 0]).max()
        rest = np.abs(k[:, 1:] - kb[:, 1:]).max()
        try:
            with np.errstate(all="ignore"):
                lo, _ = O.kbdm(sig * f, 5e-4, m=150, p=1, l=None, q=0)
            ko = canonical(lo[keep_mask(lo)])
            oref = f"oracle kept {len(ko)} finite {np.isfinite(ko).all()}"
        except Exception as e:
            oref = "oracle: " + type(e).__name__
        print(f"{f:8.0e}: kept {len(k)} amplitude dev {ra:.1e} other columns dev {rest:.1e} | {oref}")
    except Exception as e:
        print(f, "GPU:", type(e).__name__, str(e)[:100])
