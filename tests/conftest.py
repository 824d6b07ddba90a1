import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Tests marked `gpu` are skipped (not failed) on a box without a HIP device, e.g. a plain `pytest tests`
    in the build container."""
    gpu_items = [it for it in items if "gpu" in it.keywords]
    if not gpu_items:
        return
    # A missing or ABI-mismatched library is an ERROR, not a reason to skip: only "the library loads and sees no
    # device" skips.  KBDM_REQUIRE_GPU=1 (set it on a GPU runner) turns even that into a failure.
    from llckbdm_amd import _lib
    have = _lib.load().kbdm_device_count() > 0
    if not have and os.environ.get("KBDM_REQUIRE_GPU") == "1":
        raise pytest.UsageError("KBDM_REQUIRE_GPU=1 but libkbdm_hip.so sees no HIP device")
    if not have:
        skip = pytest.mark.skip(reason="no HIP device visible")
        for it in gpu_items:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    """Golden vectors produced by the reference itself (tests/golden/make_golden.py)."""
    path = os.path.join(ROOT, "tests", "golden", "kbdm_golden.npz")
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def dwell():
    return 5e-4
