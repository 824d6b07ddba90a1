"""Build libkbdm_hip.so (gfx950) in-tree with hipcc.  ``python -m llckbdm_amd.build``."""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libkbdm_hip.so")
SOURCES = ["kbdm_hip.hip"]
HEADERS = ["kbdm_kernels.hpp", "kbdm_device.h", "kb_complex.hpp", "kb_ctx.hpp", "kb_svd.hpp", "kb_eig.hpp", "kb_hqr_ms.hpp", "kb_hqr2.hpp", "kb_bdsdc.hpp", "kbdm_dc_kernels.hpp", "kb_aberth.hpp", "kbdm_ab_kernels.hpp", "kbdm_next.hpp", "kbdm_cluster.hpp", "kb_team.hpp", "kb_panel_team.hpp",
           os.path.join("..", "..", "include", "kbdm_hip.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_library(force=False, verbose=False):
    """Compile every HIP source for gfx950 into llckbdm_amd/libkbdm_hip.so."""
    if not force and not _stale():
        _record_head()
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    # -amdgpu-mfma-vgpr-form: MFMA accumulators in ordinary VGPRs (gfx90a+ register file).  Kernels whose accumulators are
    # loop-carried AND whose other phases fill the VGPR budget (k_ab_iter) otherwise copy 64 accumulator registers to the
    # AGPRs and back in every iteration of the product loop.
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wno-unused-value",
           "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    _record_head()
    return LIB


def _record_head():
    """Leave the git head of the tree the library was built from next to it (the GPU boxes get a snapshot without .git;
    bench.py puts it into its JSON line)."""
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip()
        dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "--untracked-files=no"], capture_output=True, text=True,
                               timeout=10).stdout.strip()
        if head:
            with open(os.path.join(os.path.dirname(LIB), "build_head.txt"), "w") as f:
                f.write(head + ("+dirty" if dirty else "") + "\n")
    except Exception:
        pass


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
