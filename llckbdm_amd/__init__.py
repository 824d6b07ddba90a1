"""MI355X-native KBDM ensemble solver with the call surface of danilomendesdias/llckbdm.

Module layout mirrors the reference package (``llckbdm.kbdm``, ``llckbdm.sampling``,
``llckbdm.sig_gen``): ``llckbdm_amd.kbdm.kbdm``, ``llckbdm_amd.sampling.sample_kbdm`` ...
All numerics run in hand-written HIP kernels behind ``libkbdm_hip.so`` (C ABI in
``include/kbdm_hip.h``); there is no CPU compute path in this package.
"""
__version__ = "0.1.0"
