"""CPU: the pieces of bench.py's JSON line that need no GPU - the roofline objects are built per KERNEL from the library's
per-kernel timers (`kernel` must name what `avg_ms` times), the algorithmic quantities follow SURVEY 8d."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


def test_roofline_entries_are_per_kernel_and_priced_as_documented():
    b = _bench()
    from llckbdm_amd import _lib
    ms = np.arange(100, 401, 2)
    stage_ms = {"k_svd_fac": 18.0, "k_hess": 14.0, "k_hqr": 12.0, "k_hankel": 0.03}
    kernel_ms = {"k_ab_iter": (12.0, 36), "k_bidiag_panel_team": (16.0, 10), "k_hess_panel_team": (11.0, 10),
                 "k_trail_update": (0.5, 10), "k_hess_update": (0.6, 10), "k_hankel": (0.03, 1), "k_hess_z": (1.0, 10), "k_wy_apply": (2.0, 13)}
    kern, stages = b.build_rooflines(ms, 32, stage_ms, kernel_ms)
    names = [r["kernel"] for r in kern]
    assert names[0] == "k_bidiag_panel_team" and names == sorted(names, key=lambda k: -kernel_ms[k][0])
    lane0 = sorted(ms.tolist(), reverse=True)[:32]
    for r in kern:
        tot, n = kernel_ms[r["kernel"]]
        assert r["launches"] == n and abs(r["avg_ms"] - tot / n) < 1e-12 and 0 < r["frac"] < 1
        assert r["peak"] == (b.FP64_PEAK_TFLOPS if r["bound"] == "mfma" else b.HBM_PEAK_GBS)
    ab = next(r for r in kern if r["kernel"] == "k_ab_iter")
    want = (100.0 - 56.0 / 3.0 - 16.0) * sum(float(m) ** 3 for m in lane0) / 36            # eig 100 l^3 - Hessenberg - vectors
    assert abs(ab["algorithmic_flops_per_launch"] - want) < 1e-6 * want
    tu = next(r for r in kern if r["kernel"] == "k_trail_update")
    want = sum(sum(8.0 * 64 * (m - 32 * (p + 1)) ** 2 for p in range((m - 64) // 32)) for m in lane0) / 10
    assert abs(tu["algorithmic_flops_per_launch"] - want) < 1e-6 * want
    assert stages[0]["stage"] == "k_svd_fac" and abs(stages[0]["algorithmic_flops"] - 32.0 / 3.0 * sum(float(m) ** 3 for m in lane0)) < 1.0
    # the library's kernel classes are the names the bench prices
    src = open(os.path.join(ROOT, "llckbdm_amd", "csrc", "kbdm_hip.hip")).read()
    for k in kernel_ms:
        assert f'"{k}"' in src, k
    assert _lib.KBDM_NKCLASSES == 8


def test_whole_pipeline_is_priced_with_216_m_cubed():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "pipe216 = 216.0 * float(np.sum(np.asarray(ms, dtype=np.float64) ** 3))" in src
    b = _bench()
    f = b.stage_flops(300, 300)
    assert abs(sum(f.values()) / 300.0 ** 3 - 167.0) < 1.0          # the per-stage model's own total (stated in the JSON line)
