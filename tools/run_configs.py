"""BASELINE.json configs 3-5 at FULL size on one GPU: wall time, solves/s and the size-independent properties the
parity tests use at reduced size (status words, Frobenius identity of the singular values, ordering, finite kept
lines, bit-identical results for a member solved on its own).  Usage: python tools/run_configs.py [C3 C4 C5]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from llckbdm_amd import datasets
from llckbdm_amd.engine import Engine

which = sys.argv[1:] or ["C3", "C5", "C4"]
eng = Engine(0)
out = {}
for name in which:
    if name == "C3":
        sigs, sig_idx, ms = datasets.config3()
    elif name == "C4":
        sigs, sig_idx, ms = datasets.config4()
    else:
        sigs, sig_idx, ms = datasets.config5()
    t0 = time.perf_counter()
    plan = eng.plan(sigs.shape[0], sigs.shape[1], sig_idx, ms, ms, p=1, q=0.0, dwell=datasets.DWELL)
    plan.upload(sigs)
    t1 = time.perf_counter()
    plan.execute(sync=True)
    t2 = time.perf_counter()
    res = plan.download()
    t3 = time.perf_counter()
    B = len(ms)
    bad = int((res.status & 3).sum())
    weak = int(((res.status & 4) != 0).sum())
    # Frobenius identity on a sample of members; sortedness; finite kept lines
    rng = np.random.default_rng(0)
    pick = sorted(set([0, B - 1] + list(rng.integers(0, B, 30))))
    worst = 0.0
    for i in pick:
        m = int(ms[i]); sig = sigs[sig_idx[i]]
        sv = res.singular_values(i)
        cnt = np.minimum(np.arange(2 * m - 1) + 1, 2 * m - 1 - np.arange(2 * m - 1))
        fro2 = float(np.sum(cnt * np.abs(sig[:2 * m - 1]) ** 2))
        worst = max(worst, abs(np.sum(sv ** 2) - fro2) / fro2)
        assert np.all(np.diff(sv) <= 0) and sv[-1] >= 0
        ll = res.line_list(i)
        assert np.isfinite(ll[res.keep_mask(i)]).all()
    # a member solved alone gives the same bits
    same = True
    for i in (pick[0], pick[len(pick) // 2], pick[-1]):
        solo = eng.solve(sigs[sig_idx[i]].reshape(1, -1), [0], [int(ms[i])], None, p=1, q=0.0, dwell=datasets.DWELL)
        same &= bool(np.array_equal(solo.line_list(0), res.line_list(i)))
    kept = int(sum(res.keep_mask(i).sum() for i in pick)) / len(pick)
    out[name] = {"members": B, "plan_s": t1 - t0, "execute_s": t2 - t1, "download_s": t3 - t2,
                 "solves_per_s": B / (t2 - t1), "status_fail": bad, "status_weak_eigvec": weak,
                 "max_rel_frobenius_defect": worst, "solo_bit_identical": same, "mean_kept_lines_sample": kept}
    print(json.dumps({name: out[name]}), flush=True)
    plan.close()
