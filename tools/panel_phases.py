"""GPU: where a column of the cooperative bidiagonalisation panel spends its time.  Builds a diagnostic library with
-DKB_PANEL_PROF (phase timers in kb_panel_team.hpp) into tools/_libs/, runs lane 0 of a C2 ensemble with teams of T workgroups
and prints microseconds per column and phase.  `python tools/panel_phases.py [T ...]`"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PH = ["first column + sync", "larfg v + V store", "w1 w2 dots", "column dots A0^H v", "row sweep + put", "SYNC 1", "gather y, row",
      "larfg u + U store", "z1 z2 dots", "x sweep (hp, np)", "row product A0 u", "combine + put", "SYNC 2", "gather x, column"]


def main():
    out = os.path.join(ROOT, "tools", "_libs")
    os.makedirs(out, exist_ok=True)
    lib = os.path.join(out, "libkbdm_prof.so")
    csrc = os.path.join(ROOT, "llckbdm_amd", "csrc")
    deps = [os.path.join(csrc, f) for f in os.listdir(csrc)]
    if not os.path.exists(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wno-unused-value",
                        "-DKB_PANEL_PROF", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-o", lib, os.path.join(csrc, "kbdm_hip.hip")], check=True)
    if "--build-only" in sys.argv:
        return
    os.environ["KBDM_LIB"] = lib
    from llckbdm_amd import _lib, datasets
    from llckbdm_amd.engine import Engine
    h = _lib.load()
    assert os.path.samefile(h._name, lib), h._name
    hip = ctypes.CDLL("libamdhip64.so")
    sig, _, m = datasets.config2()
    m = m[-32:]                                   # lane 0 of the C2 ensemble: m = 338..400
    for T in [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 4, 8]:
        os.environ["KBDM_PANEL_T"] = str(T)
        os.environ["KBDM_PANEL_BUDGET"] = "256"
        os.environ["KBDM_LANES"] = "1"
        eng = Engine(0, in_flight=1)
        idx = np.zeros(len(m), np.int32)
        eng.solve(sig, idx, m, dwell=5e-4)
        sym = ctypes.c_void_p()
        nbytes = ctypes.c_size_t()
        # the device symbol lives in the library's code object
        getsym = h.kbdm_debug_symbol
        getsym.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
        assert getsym(b"kb_panel_prof", ctypes.byref(sym), ctypes.byref(nbytes)) == 0
        zero = np.zeros(32, np.uint64)
        hip.hipMemcpy(sym, zero.ctypes.data_as(ctypes.c_void_p), 256, 1)
        pend = eng.submit(sig, idx, m, dwell=5e-4)
        pend.result(check=False)
        st = pend.plan.stage_ms()
        got = np.zeros(32, np.uint64)
        hip.hipMemcpy(got.ctypes.data_as(ctypes.c_void_p), sym, 256, 2)
        ncol = sum(((int(x) - 64) // 32) * 32 for x in m)           # columns in panels over the 32 members
        print("T=%d: k_svd_fac %.2f ms, %d panel columns over 32 members" % (T, st["k_svd_fac"], ncol))
        tot = 0.0
        for i, name in enumerate(PH):
            us = float(got[i]) / 100.0 / ncol
            tot += us
            print("   %-22s %7.2f us / column" % (name, us))
        print("   %-22s %7.2f us / column" % ("total", tot), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
