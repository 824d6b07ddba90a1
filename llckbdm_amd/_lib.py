"""ctypes binding of libkbdm_hip.so (C ABI: include/kbdm_hip.h).

Fails loudly when the HIP library is missing or no GPU is visible: this package has no
CPU compute path (the numpy oracle under ``oracle/`` is test infrastructure and is never
imported from here).
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint8, c_void_p

_PKG = os.path.dirname(os.path.abspath(__file__))
# (KBDM_LIB: a diagnostic build of the same sources, e.g. tools/panel_phases.py's library with phase timers)
LIB_PATH = os.environ.get("KBDM_LIB") or os.path.join(_PKG, "libkbdm_hip.so")

# A plan runs on three HIP streams (two lanes + the critical lane's side stream) and an Engine keeps up to three
# contexts in flight.  The ROCm runtime maps streams onto four hardware queues by default; streams that share a queue
# serialise (correct, but the lanes then no longer overlap).  Effective only if HIP has not been initialised in this
# process yet; never overrides the user's own setting.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

KBDM_ABI_VERSION = 3
KBDM_NSTAGES = 16
KBDM_UNIQUE_ID_BYTES = 128
STAT_SVD_NOCONV, STAT_EIG_NOCONV, STAT_INVIT_WEAK = 1, 2, 4
MODE_SOLO_QR = 2
MODE_KERNEL_TIMERS = 4
KBDM_NKCLASSES = 8

# every symbol include/kbdm_hip.h declares: (restype, argtypes)
_P = c_void_p
SYMBOLS = {
    "kbdm_abi_version": (c_int, []),
    "kbdm_device_count": (c_int, []),
    "kbdm_last_error": (c_char_p, []),
    "kbdm_ctx_create": (c_int, [c_int, POINTER(_P)]),
    "kbdm_ctx_create_lanes": (c_int, [c_int, c_int, POINTER(_P)]),
    "kbdm_ctx_destroy": (c_int, [_P]),
    "kbdm_ctx_set_panel_teams": (c_int, [_P, c_int, c_int, c_int, c_double]),
    "kbdm_plan_create": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, c_int, c_double, c_double, POINTER(_P)]),
    "kbdm_plan_destroy": (c_int, [_P]),
    "kbdm_plan_total_lines": (c_int64, [_P]),
    "kbdm_plan_total_sv": (c_int64, [_P]),
    "kbdm_plan_offsets": (c_int, [_P, _P, _P]),
    "kbdm_plan_upload": (c_int, [_P, _P]),
    "kbdm_plan_execute": (c_int, [_P]),
    "kbdm_plan_wait_stage": (c_int, [_P, c_int]),
    "kbdm_plan_sync": (c_int, [_P]),
    "kbdm_plan_download": (c_int, [_P, _P, _P, _P, _P, _P]),
    "kbdm_plan_submit": (c_int, [_P, _P]),
    "kbdm_plan_collect": (c_int, [_P, _P, _P, _P, _P, _P]),
    "kbdm_plan_set_mode": (c_int, [_P, c_int]),
    "kbdm_debug_force_status": (c_int, [c_int, c_int]),
    "kbdm_kernel_class_name": (c_char_p, [c_int]),
    "kbdm_plan_kernel_ms": (c_int, [_P, c_int, _P, _P]),
    "kbdm_workspace_estimate": (c_int64, [c_int, _P, _P]),
    "kbdm_plan_workspace_bytes": (c_int64, [_P]),
    "kbdm_plan_lines_device": (_P, [_P]),
    "kbdm_plan_sv_device": (_P, [_P]),
    "kbdm_plan_copy_lines_device": (c_int, [_P, _P, c_int64]),
    "kbdm_plan_stage_ms": (c_int, [_P, _P, c_int]),
    "kbdm_stage_name": (c_char_p, [c_int]),
    "kbdm_plan_lane0_members": (c_int, [_P]),
    "kbdm_plan_eig_fallbacks": (c_int, [_P]),
    "kbdm_ctx_last_eig_fallbacks": (c_int, [_P]),
    "kbdm_plan_ab_stats": (c_int, [_P, _P, c_int]),
    "kbdm_rmse_batch": (c_int, [_P, _P, c_int, c_double, _P, _P, c_int, _P]),
    "kbdm_silhouette_samples": (c_int, [_P, _P, c_int, c_int, _P, _P]),
    "kbdm_silhouette_sweep": (c_int, [_P, _P, c_int, c_int, _P, c_int, _P, _P]),
    "kbdm_hdbscan_sweep": (c_int, [_P, _P, c_int, c_int, _P, c_int, c_int, _P, _P]),
    "kbdm_core_distances": (c_int, [_P, _P, c_int, c_int, _P, c_int, _P]),
    "kbdm_hdbscan_labels_from_mst": (c_int, [c_int, _P, _P, _P, c_int, _P]),
    "kbdm_comm_unique_id": (c_int, [_P]),
    "kbdm_comm_init": (c_int, [_P, c_int, c_int, _P]),
    "kbdm_comm_destroy": (c_int, [_P]),
    "kbdm_comm_attach": (c_int, [_P, _P]),
    "kbdm_packed_bytes": (c_int64, [c_int64, c_int64, c_int64]),
    "kbdm_plan_gather": (c_int, [_P, c_int, c_int, _P, c_int, _P]),
    "kbdm_gathered_device": (_P, [_P]),
    "kbdm_gather_wait": (c_int, [_P]),
    "kbdm_solve_batch": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, c_int, c_double, c_double,
                                 _P, _P, _P, _P, _P]),
    "kbdm_hankel_batch": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, c_int, _P, _P, _P]),
    "kbdm_svd_batch": (c_int, [_P, _P, c_int, _P, _P, _P, _P, _P]),
    "kbdm_eig_batch": (c_int, [_P, _P, c_int, _P, _P, _P, _P]),
}

_lib = None


class KbdmHipError(RuntimeError):
    """A libkbdm_hip.so call returned a negative KBDM_E_* code."""


def load():
    """Load the shared library and bind every declared symbol (no GPU call is made)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m llckbdm_amd.build` (hipcc, gfx950). "
            "llckbdm_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)   # AttributeError here = ABI mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.kbdm_abi_version() != KBDM_ABI_VERSION:
        raise ImportError(f"libkbdm_hip.so ABI {lib.kbdm_abi_version()} != expected {KBDM_ABI_VERSION}")
    _lib = lib
    return lib


def check(code):
    if code != 0:
        msg = load().kbdm_last_error()
        raise KbdmHipError(f"libkbdm_hip error {code}: {msg.decode() if msg else ''}")


def ptr(arr):
    """Raw pointer of a C-contiguous numpy array (or None)."""
    return None if arr is None else arr.ctypes.data_as(c_void_p)
