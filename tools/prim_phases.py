"""GPU: where a step of k_prim_mst_reg spends its time (diagnostic library with -DKB_PANEL_PROF, as tools/panel_phases.py)."""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "tools", "_libs", "libkbdm_prof.so")
if "--build-only" in sys.argv:
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    csrc = os.path.join(ROOT, "llckbdm_amd", "csrc")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wno-unused-value",
                    "-DKB_PANEL_PROF", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-o", lib, os.path.join(csrc, "kbdm_hip.hip")], check=True)
    sys.exit(0)
os.environ["KBDM_LIB"] = lib
from llckbdm_amd import _lib
from llckbdm_amd.engine import Engine
h = _lib.load()
hip = ctypes.CDLL("libamdhip64.so")
rng = np.random.default_rng(0)
n = 20000
X = rng.standard_normal((n, 4))
eng = Engine(0)
eng.hdbscan_sweep(X, [1])
sym, nb = ctypes.c_void_p(), ctypes.c_size_t()
g = h.kbdm_debug_symbol
g.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
assert g(b"kb_panel_prof", ctypes.byref(sym), ctypes.byref(nb)) == 0
for ks in ([1], [1, 50, 150]):
    z = np.zeros(32, np.uint64)
    hip.hipMemcpy(sym, z.ctypes.data_as(ctypes.c_void_p), 256, 1)
    eng.hdbscan_sweep(X, ks)
    got = np.zeros(32, np.uint64)
    hip.hipMemcpy(got.ctypes.data_as(ctypes.c_void_p), sym, 256, 2)
    names = ["relaxation", "wave argmin + LDS", "barrier B", "final argmin", "winner's tail", "barrier A"]
    tot = 0.0
    print("fits", ks)
    for i, nm in enumerate(names):
        us = float(got[20 + i]) / 100.0 / (n - 1)
        tot += us
        print("   %-20s %6.2f us / step" % (nm, us))
    print("   %-20s %6.2f us / step" % ("total", tot))
