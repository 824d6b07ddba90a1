"""GPU: a few C2 ensembles, one at a time (for rocprofv3 runs).  Environment knobs as usual."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llckbdm_amd import datasets
from llckbdm_amd.engine import Engine
sig, idx, m = datasets.config2()
eng = Engine(0, in_flight=1)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    eng.solve(sig, idx, m, dwell=5e-4)
eng.close()
