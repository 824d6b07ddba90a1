"""GPU: time of Engine.hdbscan_sweep / silhouette_samples on the pooled lines of a C2 ensemble, by number of fits."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from llckbdm_amd import datasets
from llckbdm_amd.engine import Engine
from llckbdm_amd.sampling import sample_kbdm, filter_samples
from llckbdm_amd.llckbdm import _transform_line_lists
sig, idx, m = datasets.config2(seed=0)
sig = np.atleast_2d(sig)[0]
eng = Engine(0)
ll, _ = sample_kbdm(sig, datasets.DWELL, [int(x) for x in m], p=1, l=None, q=0, engine=eng)
X = _transform_line_lists(filter_samples(np.concatenate(ll)), datasets.DWELL)
print("points", X.shape)
for ks in ([1], [150], list(range(1, 16)), list(range(1, 151))):
    eng.hdbscan_sweep(X, ks)
    t0 = time.perf_counter(); lab, ncl = eng.hdbscan_sweep(X, ks); dt = time.perf_counter() - t0
    print("fits %3d (k up to %3d): %.3f s" % (len(ks), max(ks), dt), flush=True)
t0 = time.perf_counter(); s = eng.silhouette_samples(X, lab[10]); print("one silhouette call: %.4f s" % (time.perf_counter() - t0))
