"""GPU: the Hankel build alone (kbdm_hankel_batch, one output, 384 members of m = 512: nothing else on the GPU) - run
under rocprofv3 --kernel-trace (tools/hankel_bw.sh) and read k_hankel's duration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from llckbdm_amd import _lib
from llckbdm_amd.engine import Engine
eng = Engine(0, in_flight=1)
rng = np.random.default_rng(0)
B, M, N = 384, 512, 2048
sig = (rng.standard_normal((1, N)) + 1j * rng.standard_normal((1, N)))
m = np.full(B, M, np.int32); idx = np.zeros(B, np.int32)
out = np.empty(B * M * M, np.complex128)
for var in range(6):
    _lib.check(eng.lib.kbdm_hankel_batch(eng.ctx, _lib.ptr(sig), 1, N, B, _lib.ptr(idx), _lib.ptr(m), 1, _lib.ptr(out), None, None))
    ref = sig[0][np.add.outer(np.arange(M), np.arange(M))]
    assert np.array_equal(out[:M * M].reshape(M, M), ref) and np.array_equal(out[-M * M:].reshape(M, M), ref), var
print("ok")
