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
    try:
        from llckbdm_amd import _lib
        have = _lib.load().kbdm_device_count() > 0
    except Exception:
        have = False
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
