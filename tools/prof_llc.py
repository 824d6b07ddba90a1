"""GPU: where llc_kbdm on the C2 ensemble spends its time (cProfile, cumulative)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from llckbdm_amd import datasets
from llckbdm_amd.llckbdm import llc_kbdm
sig, idx, m = datasets.config2(seed=0)
sig = np.atleast_2d(sig)[0]
m_range = [int(x) for x in m]
llc_kbdm(sig, datasets.DWELL, m_range, p=1, l=None)
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
res = llc_kbdm(sig, datasets.DWELL, m_range, p=1, l=None)
pr.disable()
print("llc_kbdm: %.3f s, %d lines" % (time.perf_counter() - t0, len(res.line_list)))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
