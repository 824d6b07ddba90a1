"""Random parity sweep on the GPU box: many ragged members (random m, l, noise seed; N = 1024 or 2048) against the
oracle run on the host, reporting the worst errors instead of asserting.  Checker use of the oracle only.
usage: python tools/stress_parity.py [n_members] [seed] [p] [q]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import kbdm_oracle as O
from llckbdm_amd.engine import Engine
from tests.helpers import canonical, keep_mask

B = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
PP = int(sys.argv[3]) if len(sys.argv) > 3 else 1
QQ = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
rng = np.random.default_rng(seed)
DW = 5e-4
base = {1024: O.brain_sim_signal(1024), 2048: O.brain_sim_signal(2048)}
sigs, Ns, ms, ls = [], [], [], []
for b in range(B):
    N = int(rng.choice([1024, 2048]))
    sigma = float(rng.choice([1e-3, 1e-2, 1e-4]))
    m = int(rng.integers(20, 330))
    l = m if rng.random() < 0.7 else int(rng.integers(max(4, m // 4), m + 1))
    sigs.append(O.make_noisy(base[N], sigma, 7000 + seed * 1000 + b)); Ns.append(N); ms.append(m); ls.append(l)
eng = Engine(0)
t0 = time.time()
res = {}
for N in (1024, 2048):
    idx = [i for i in range(B) if Ns[i] == N]
    if not idx:
        continue
    S = np.stack([sigs[i] for i in idx])
    r = eng.solve(S, list(range(len(idx))), [ms[i] for i in idx], [ls[i] for i in idx], p=PP, q=QQ, dwell=DW)
    for j, i in enumerate(idx):
        res[i] = (r.line_list(j).copy(), r.keep_mask(j).copy(), r.singular_values(j).copy(), int(r.status[j]))
tg = time.time() - t0
worst = {"sv": 0.0, "strong_rel": 0.0, "all_rel": 0.0, "phase": 0.0}
count_mismatch, status_bad, weak = [], [], 0
t0 = time.time()
for i in range(B):
    got, keep, sv, st = res[i]
    want, info = O.kbdm(sigs[i], DW, m=ms[i], l=ls[i], p=PP, q=QQ, normalizer="gemm")
    if st & 3: status_bad.append((i, ms[i], ls[i], st))
    weak += 1 if st & 4 else 0
    worst["sv"] = max(worst["sv"], float(np.abs(sv - info.singular_values).max() / (info.singular_values[0] * ms[i])))
    k, w = canonical(got[keep_mask(got)]), canonical(O.filter_samples(want))
    if len(k) != len(w) or not np.array_equal(keep, keep_mask(got)):
        count_mismatch.append((i, ms[i], ls[i], len(k), len(w)))
        continue
    if len(k) == 0:
        continue
    rel = np.abs(k[:, :3] - w[:, :3]) / np.maximum(np.abs(w[:, :3]), 1e-300)
    dph = np.abs(np.angle(np.exp(1j * (k[:, 3] - w[:, 3]))))
    strong = w[:, 0] > 1e-4
    if strong.any():
        worst["strong_rel"] = max(worst["strong_rel"], float(rel[strong].max()))
        worst["phase"] = max(worst["phase"], float(dph[strong].max()))
    worst["all_rel"] = max(worst["all_rel"], float(rel.max()))
print(json.dumps({"members": B, "seed": seed, "p": PP, "q": QQ, "gpu_s": tg, "oracle_s": time.time() - t0, "worst": worst,
                  "kept_count_mismatch": count_mismatch, "status_fail": status_bad, "status_weak_eigvec": weak}))
