"""CPU: the C-ABI library builds, loads and exports every symbol include/kbdm_hip.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "kbdm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kbdm_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from llckbdm_amd import build, _lib
    build.build_library()
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in kbdm_hip.h but not exported"
    # and the binding table covers exactly the header
    assert sorted(_lib.SYMBOLS) == declared
    assert lib.kbdm_abi_version() == _lib.KBDM_ABI_VERSION


def test_stage_names():
    from llckbdm_amd import _lib
    lib = _lib.load()
    names = [lib.kbdm_stage_name(i).decode() for i in range(_lib.KBDM_NSTAGES)]
    assert names[0] == "k_hankel" and names[-1] == "k_epilogue" and len(set(names)) == _lib.KBDM_NSTAGES


def test_no_cpu_fallback_without_device():
    """Without a GPU the product must fail loudly, never compute on the CPU."""
    from llckbdm_amd import _lib
    lib = _lib.load()
    if lib.kbdm_device_count() > 0:
        pytest.skip("a GPU is visible")
    import numpy as np
    from llckbdm_amd.kbdm import kbdm
    with pytest.raises(_lib.KbdmHipError):
        kbdm(np.ones(64, dtype=complex), 1e-3, m=8)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "llckbdm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+\.*oracle", src, flags=re.M), f
                assert "hostsim" not in src and "import scipy" not in src and "from scipy" not in src, f
