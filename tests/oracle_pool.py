"""Test helper (run as a child process by GPU tests): the oracle's filtered line lists of a whole ensemble, one host
process per core.  A separate interpreter because a process that holds a HIP context must not fork.
    python tests/oracle_pool.py <signal.npy> <ms.npy> <out.npz>"""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
DWELL = 5e-4


def _member(args):
    sig, m = args
    from oracle import kbdm_oracle as O
    ll, _ = O.kbdm(sig, DWELL, m=int(m), normalizer="gemm")
    return O.filter_samples(ll)


if __name__ == "__main__":
    os.environ["OPENBLAS_NUM_THREADS"] = "1"
    sig, ms, out = np.load(sys.argv[1]), np.load(sys.argv[2]), sys.argv[3]
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_member, [(sig, m) for m in ms[::-1]], chunksize=1)[::-1]
    np.savez(out, **{f"m{int(m)}": r for m, r in zip(ms, res)})
    print("oracle members done:", len(res), flush=True)
