"""CPU (-m "not gpu"): the rows next to the hot path - the oracle against the vectors the
reference produced (tests/golden/next_golden.npz, make_golden_next.py), the known-answer checks of the
reference's own tests for the pure-numpy helpers, and the host logic of llc_kbdm / min_rmse_kbdm driven by an
oracle-backed engine."""
import os

import numpy as np
import pytest

from oracle import llc_oracle as L
from tests.fake_engine import OracleEngine

DWELL = 5e-4
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gnext():
    return np.load(os.path.join(HERE, "golden", "next_golden.npz"))


def test_oracle_rmse_matches_reference_vectors(gnext):
    for i in range(int(gnext["rmse_ncand"][0])):
        c = gnext[f"rmse_cand{i}"]
        assert L.calculate_freq_domain_rmse(gnext["sig2048"], c, DWELL) == pytest.approx(gnext[f"rmse_clean{i}"][0], rel=1e-13, abs=1e-15)
        assert L.calculate_freq_domain_rmse(gnext["noisy2048"], c, DWELL) == pytest.approx(gnext[f"rmse_noisy{i}"][0], rel=1e-13, abs=1e-15)
    assert L.calculate_freq_domain_rmse(gnext["sig1000"], gnext["rmse_cand3"], DWELL) == pytest.approx(gnext["rmse_odd"][0], rel=1e-13)


def test_time_domain_identity_used_by_the_kernel(gnext):
    """rmse^2 = (sum |r|^2 + Re sum r_n r_{(N-n) mod N}) / (2N): what k_rmse evaluates (kbdm_next.hpp)."""
    data, c = gnext["noisy2048"], gnext["rmse_cand4"]
    N = len(data)
    r = data - L.multi_fid(np.arange(N) * DWELL, c)
    v = np.sqrt((np.sum(np.abs(r) ** 2) + np.real(np.sum(r * r[(-np.arange(N)) % N]))) / (2 * N))
    assert v == pytest.approx(gnext["rmse_noisy4"][0], rel=1e-13)


def test_transform_known_answers(gnext):
    """The reference's own checks: _tests/test_llckbdm.py:11-37."""
    from llckbdm_amd.llckbdm import _transform_line_lists, _inverse_transform_line_lists
    params = gnext["params"]
    for transform, inverse in ((_transform_line_lists, _inverse_transform_line_lists),
                               (L.transform_line_lists, L.inverse_transform_line_lists)):
        t = transform(params, DWELL)
        MU = t[:, 0] + 1j * t[:, 1]
        OMEGA = np.log(MU) / (1j * DWELL)
        assert t[:, 2] == pytest.approx(params[:, 0])
        assert 1. / np.imag(OMEGA) == pytest.approx(params[:, 1])
        assert np.real(OMEGA) / (2 * np.pi) == pytest.approx(params[:, 2])
        assert t[:, 3] == pytest.approx(params[:, 3])                 # the fixture's phases are 0 (and the feature is zeroed)
        assert params == pytest.approx(inverse(t, DWELL))
    assert np.array_equal(_transform_line_lists(params, DWELL), L.transform_line_lists(params, DWELL))


def test_summarize_clusters_is_harmonic_in_t2():
    from llckbdm_amd.llckbdm import _summarize_clusters
    samples = np.array([[1.0, 0.1, 10.0, 0.0], [3.0, 0.3, 12.0, 0.2], [5.0, 0.2, 50.0, 0.0], [7.0, 0.2, 52.0, 0.4]])
    clusters = [np.nonzero(np.array([1, 1, 0, 0])), np.nonzero(np.array([0, 0, 1, 1]))]
    before = samples.copy()
    out = _summarize_clusters(samples, clusters)
    assert np.array_equal(samples, before)                            # the caller's pooled samples are not modified
    assert out[0] == pytest.approx([2.0, 2.0 / (1 / 0.1 + 1 / 0.3), 11.0, 0.1])
    assert out[1] == pytest.approx([6.0, 0.2, 51.0, 0.2])
    assert out == pytest.approx(L.summarize_clusters(samples, clusters))


def test_silhouette_restatement_against_sklearn():
    from sklearn.metrics import silhouette_samples
    rng = np.random.default_rng(3)
    X = np.concatenate([rng.standard_normal((40, 4)) * 0.05 + c for c in ([0, 0, 0, 0], [1, 0, 0.5, 0], [0, 2, 0, 0])] +
                       [rng.standard_normal((15, 4))])
    labels = np.array([0] * 40 + [1] * 40 + [2] * 40 + [-1] * 14 + [3])       # noise is a class, one singleton
    assert np.abs(L.silhouette_samples_direct(X, labels) - silhouette_samples(X, labels)).max() < 1e-12


def test_min_rmse_kbdm_host_logic(gnext):
    from llckbdm_amd.min_rmse_kbdm import min_rmse_kbdm
    eng = OracleEngine()
    sig = gnext["sig2048"]
    res = min_rmse_kbdm(data=sig, dwell=DWELL, m_range=[int(m) for m in gnext["minrmse_m_range"]], l=30, engine=eng)
    assert len(res.samples) == 6                                        # _tests/test_min_rmse_kbdm.py:18
    assert res.min_rmse == pytest.approx(0, abs=1e-10) and res.min_index == 2
    assert res.line_list is res.samples[2] and len(res.rmses_list) == 6
    # empty candidates score inf, no candidates -> None (min_rmse_kbdm.py:36-37, 55)
    res = min_rmse_kbdm(data=sig, dwell=DWELL, samples=[np.zeros((0, 4)), gnext["params"]], engine=eng)
    assert res.rmses_list[0] == np.inf and res.min_index == 1
    assert min_rmse_kbdm(data=sig, dwell=DWELL, samples=[], engine=eng) is None


def test_llc_kbdm_host_logic(gnext):
    """Reference _tests/test_llckbdm.py:40-69 with the oracle as the numerical back end."""
    from llckbdm_amd.llckbdm import llc_kbdm, LlcKbdmResult
    from llckbdm_amd.sampling import filter_samples
    eng = OracleEngine()
    sig, params = gnext["sig2048"], gnext["params"]
    with pytest.raises(ValueError) as e:
        llc_kbdm(data=sig, dwell=DWELL, m_range=[1], engine=eng)
    assert "size of 'm_range' must be greater than 2" in str(e.value)
    results = llc_kbdm(data=sig, dwell=DWELL, m_range=range(250, 256), p=1, l=30, engine=eng)
    assert isinstance(results, LlcKbdmResult)
    line_list = filter_samples(results.line_list, amplitude_tol=1e-3)
    assert len(line_list) == len(params)
    est = L.multi_fid(np.arange(len(sig)) * DWELL, line_list)
    assert np.std(est.real - sig.real) < 1e-3 and np.std(est.imag - sig.imag) < 1e-3
    assert len(results.silhouette) == len(results.line_list) and results.rmse < 1e-6


def test_llc_kbdm_has_no_silent_host_clusterer(gnext, monkeypatch):
    """with clusterer='gpu' EVERY fit of the sweep goes to the engine (the k-nearest-neighbour kernel has no
    limit on min_samples since round 4): one batched call, no per-fit host path."""
    import llckbdm_amd.llckbdm as M
    seen = []

    class Eng(OracleEngine):
        def hdbscan_sweep(self, X, ks, mcs=5):
            seen.append(list(ks))
            return super().hdbscan_sweep(X, ks, mcs)
    sig = gnext["sig2048"][:512]                       # (control flow only: small members keep the oracle cheap)

    def never(*a, **k):
        raise AssertionError("a per-fit host clusterer ran although clusterer='gpu'")
    monkeypatch.setattr(M, "_fit_labels", never)
    res = M.llc_kbdm(data=sig, dwell=DWELL, m_range=range(60, 66), p=1, l=24, engine=Eng())
    assert len(res.line_list) > 0
    assert seen == [[1, 2, 3, 4, 5]]                   # the whole sweep (llckbdm.py:104), one batched call
    assert not hasattr(M, "GPU_SWEEP_MAX_K")


def test_hdbscan_tree_part_against_sklearn():
    """The host half of the built-in HDBSCAN* (C ABI kbdm_hdbscan_labels_from_mst, no GPU needed): fed with a
    numpy Prim MST of the mutual-reachability graph it reproduces scikit-learn's partitions (same number of
    clusters; the same samples but for a few on mutual-reachability ties)."""
    from sklearn.cluster import HDBSCAN
    from sklearn.metrics import adjusted_rand_score
    from llckbdm_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(1)
    c = rng.standard_normal((6, 4)) * 3
    lab = rng.integers(0, 6, 400)
    X = np.concatenate([c[lab] + 0.1 * rng.standard_normal((400, 4)), rng.uniform(-8, 8, (60, 4))])
    n = len(X)
    D = np.sqrt(((X[:, None, :] - X[None, :, :]) ** 2).sum(-1))
    for k in (1, 3, 10, 40):
        core = np.sort(D, axis=1)[:, k - 1]
        M = np.maximum(np.maximum(core[:, None], core[None, :]), D)
        intree = np.zeros(n, bool)
        intree[0] = True
        best, src = M[0].copy(), np.zeros(n, int)
        a, b, w = [], [], []
        for _ in range(n - 1):
            cand = np.where(intree, np.inf, best)
            j = int(np.argmin(cand))
            a.append(src[j]); b.append(j); w.append(cand[j])
            intree[j] = True
            upd = (M[j] < best) & ~intree
            src[upd] = j
            best = np.minimum(best, M[j])
        a, b, w = np.array(a, np.int32), np.array(b, np.int32), np.array(w)
        out = np.zeros(n, np.int32)
        ncl = lib.kbdm_hdbscan_labels_from_mst(n, _lib.ptr(a), _lib.ptr(b), _lib.ptr(w), 5, _lib.ptr(out))
        ref = HDBSCAN(min_samples=k, min_cluster_size=5, copy=True).fit(X).labels_
        assert ncl == len(set(ref.tolist()) - {-1})
        assert adjusted_rand_score(ref, out) >= 0.99


def test_summarize_clusters_fast_path_against_the_loop():
    """The segmented-sum form of `_summarize_clusters` (all clusters at once) against the reference's loop of
    `np.average(rows, axis=0)` per cluster: the same means to a few ulp of the summands (another order of the additions), empty clusters
    (NaN rows) included."""
    import llckbdm_amd.llckbdm as M
    rng = np.random.default_rng(3)
    samples = np.column_stack([rng.random(5000) + 0.1, rng.random(5000) + 0.01, rng.standard_normal(5000), rng.standard_normal(5000)])
    perm = rng.permutation(5000)
    cuts = np.sort(rng.choice(np.arange(1, 5000), 60, replace=False))
    clusters = [(np.sort(ix),) for ix in np.split(perm, cuts)]
    clusters.insert(7, (np.array([], dtype=np.int64),))
    clusters.append((np.array([4], dtype=np.int64),))
    fast = M._summarize_clusters(samples, clusters)
    slow = M._summarize_clusters(samples, clusters, summarizer=lambda rows, axis: np.average(rows, axis=axis))
    # (ulps of the summands: the F / PH means sit near zero)
    np.testing.assert_allclose(fast, slow, rtol=1e-14, atol=4 * np.finfo(float).eps, equal_nan=True)
    assert np.isnan(fast[7]).all() and not np.isnan(np.delete(fast, 7, axis=0)).any()
