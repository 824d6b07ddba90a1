"""GPU (-m gpu): the HDBSCAN* sweep (C ABI kbdm_hdbscan_sweep: k-nearest-neighbour distances and Prim MSTs on the
GPU, condensed trees on the host) against scikit-learn's HDBSCAN run on the same points: the same number of
clusters and the same partition up to the handful of border samples that mutual-reachability TIES hand to a
different side (adjusted Rand index >= 0.99; the reference pins no labels)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from llckbdm_amd.engine import Engine
    return Engine(0)


def _data(rng, n, nc, spread, noise):
    c = rng.standard_normal((nc, 4)) * 3
    lab = rng.integers(0, nc, n)
    X = c[lab] + spread * rng.standard_normal((n, 4)) * rng.uniform(0.2, 1, (nc, 1))[lab]
    return np.concatenate([X, rng.uniform(-8, 8, (noise, 4))])


def test_sweep_matches_sklearn(eng):
    from sklearn.cluster import HDBSCAN
    from sklearn.metrics import adjusted_rand_score
    rng = np.random.default_rng(1)
    for n, nc, spread, noise in ((300, 4, 0.1, 30), (1000, 12, 0.2, 200), (500, 3, 0.5, 0)):
        X = _data(rng, n, nc, spread, noise)
        ks = [1, 2, 5, 10, 25, 60]
        labels, ncl = eng.hdbscan_sweep(X, ks)
        for f, k in enumerate(ks):
            ref = HDBSCAN(min_samples=k, min_cluster_size=5, copy=True).fit(X).labels_
            assert ncl[f] == len(set(ref.tolist()) - {-1}) == len(set(labels[f].tolist()) - {-1})
            assert adjusted_rand_score(ref, labels[f]) >= 0.99, (n, k)
            assert abs(int((ref == -1).sum()) - int((labels[f] == -1).sum())) <= max(3, len(X) // 200)


def _canon(labels):
    """Relabel clusters in order of first appearance (noise stays -1): equal partitions <=> equal arrays."""
    out, seen = np.full(len(labels), -1, dtype=np.int64), {}
    for i, v in enumerate(labels):
        if v >= 0:
            out[i] = seen.setdefault(int(v), len(seen))
    return out


def test_tie_free_inputs_give_scikit_learns_labels_exactly(eng):
    """With min_samples <= 2 the mutual-reachability distance IS the Euclidean distance (a sample's core distance is
    at most its distance to any other sample), so random points have no ties, the minimum spanning tree and the
    single-linkage hierarchy are unique, and the labels must equal scikit-learn's - as arrays, not up to a Rand index."""
    from sklearn.cluster import HDBSCAN
    rng = np.random.default_rng(5)
    for n, nc, spread, noise in ((300, 4, 0.1, 30), (1500, 12, 0.2, 300), (700, 3, 0.5, 0), (2500, 30, 0.15, 500)):
        X = _data(rng, n, nc, spread, noise)
        labels, ncl = eng.hdbscan_sweep(X, [1, 2])
        for f, k in enumerate((1, 2)):
            ref = HDBSCAN(min_samples=k, min_cluster_size=5, copy=True).fit(X).labels_
            assert np.array_equal(_canon(labels[f]), _canon(ref)), (n, k)
            assert np.array_equal(labels[f], ref), (n, k)          # the numbering convention matches as well


def _prim_reference(X, k):
    """The documented tie rule, restated in numpy: core distances from a full sort, Prim from sample 0 over the
    mutual-reachability graph, strict '<' relaxation, the closest outside sample with the LOWEST index next."""
    n = len(X)
    D = np.sqrt(((X[:, None, :] - X[None, :, :]) ** 2).sum(-1))
    core = np.sort(D, axis=1)[:, k - 1]
    best, src, intree = np.full(n, np.inf), np.zeros(n, dtype=np.int32), np.zeros(n, dtype=bool)
    intree[0] = True
    cur, edges = 0, []
    for _ in range(n - 1):
        w = np.maximum(D[cur], np.maximum(core, core[cur]))
        upd = (~intree) & (w < best)
        best[upd], src[upd] = w[upd], cur
        cand = np.where(intree, np.inf, best)
        nxt = int(np.argmin(cand))                                # argmin: first (lowest) index among ties
        edges.append((int(src[nxt]), nxt, float(best[nxt])))
        intree[nxt] = True
        cur = nxt
    return edges


def test_documented_tie_rule_on_inputs_full_of_ties(eng):
    """A lattice with duplicated points: almost every mutual-reachability weight is tied.  The GPU sweep must follow
    the rule kbdm_cluster.hpp documents (Prim from sample 0, lowest index among equal candidates, dendrogram edges in
    stable weight order): its labels equal the host tree code run on a numpy Prim that implements exactly that rule -
    and they are the same on every run."""
    from llckbdm_amd import _lib
    g = np.stack(np.meshgrid(np.arange(6.0), np.arange(6.0), np.arange(3.0), [0.0]), -1).reshape(-1, 4)
    X = np.concatenate([g, g[:40] + 10.0, g[:15]])
    ks = [1, 3, 6]
    labels, _ = eng.hdbscan_sweep(X, ks)
    again, _ = eng.hdbscan_sweep(X, ks)
    assert np.array_equal(labels, again)
    lib = _lib.load()
    for f, k in enumerate(ks):
        e = _prim_reference(X, k)
        a = np.array([x[0] for x in e], dtype=np.int32)
        b = np.array([x[1] for x in e], dtype=np.int32)
        w = np.array([x[2] for x in e], dtype=np.float64)
        ref = np.empty(len(X), dtype=np.int32)
        lib.kbdm_hdbscan_labels_from_mst(len(X), _lib.ptr(a), _lib.ptr(b), _lib.ptr(w), 5, _lib.ptr(ref))
        assert np.array_equal(labels[f], ref), k


def test_large_min_samples_stay_on_the_gpu(eng):
    """min_samples beyond 300 (an m_range of more than 301 members, e.g. BASELINE config 4's 1001): the k-nearest-
    neighbour pass shrinks its workgroups instead of leaving the GPU; core distances checked through the labels of
    the documented rule."""
    from llckbdm_amd import _lib
    rng = np.random.default_rng(8)
    X = _data(rng, 1400, 3, 0.3, 100)
    ks = [301, 700, 1000]
    labels, ncl = eng.hdbscan_sweep(X, ks)
    lib = _lib.load()
    for f, k in enumerate(ks):
        e = _prim_reference(X, k)
        a = np.array([x[0] for x in e], dtype=np.int32)
        b = np.array([x[1] for x in e], dtype=np.int32)
        w = np.array([x[2] for x in e], dtype=np.float64)
        ref = np.empty(len(X), dtype=np.int32)
        lib.kbdm_hdbscan_labels_from_mst(len(X), _lib.ptr(a), _lib.ptr(b), _lib.ptr(w), 5, _lib.ptr(ref))
        assert np.array_equal(labels[f], ref), k


def test_sweep_degenerate_inputs(eng):
    rng = np.random.default_rng(2)
    X = rng.standard_normal((40, 4))
    labels, ncl = eng.hdbscan_sweep(X, [1, 40])
    assert labels.shape == (2, 40) and (labels >= -1).all()
    dup = np.repeat(rng.standard_normal((3, 4)), 20, axis=0)          # exact duplicates: infinite lambdas
    labels, ncl = eng.hdbscan_sweep(dup, [2, 5])
    assert ncl[0] == 3 and len(set(labels[0].tolist())) == 3
    with pytest.raises(Exception):
        eng.hdbscan_sweep(X, [0])
    with pytest.raises(Exception):
        eng.hdbscan_sweep(X, [41])


def test_core_distances_in_passes_bit_exact_on_a_lattice_full_of_ties(eng):
    """min_samples beyond one LDS list (about 2550 at 8 samples per workgroup): the k-nearest-neighbour kernel runs in
    passes that continue each other in (distance, index) order.  Integer lattice points with duplicates: every squared
    distance is a small integer, so sqrt is exact on both sides and whole runs of equal distances straddle the pass
    boundaries.  Bit-exact against a full numpy sort, for one, two and three passes."""
    rng = np.random.default_rng(12)
    X = rng.integers(0, 7, (5300, 4)).astype(np.float64)
    D = np.sqrt(((X[:, None, :] - X[None, :, :]) ** 2).sum(-1))
    D.sort(axis=1)
    ks = [1, 2, 300, 2551, 2552, 2600, 5102, 5103, 5299, 5300]
    got = eng.core_distances(X, ks)
    for f, k in enumerate(ks):
        assert np.array_equal(got[f], D[:, k - 1]), k
    sub = eng.core_distances(X[:2700], [2650, 2700])                    # two passes, the last one short
    D2 = np.sort(np.sqrt(((X[:2700, None, :] - X[None, :2700, :]) ** 2).sum(-1)), axis=1)
    assert np.array_equal(sub[0], D2[:, 2649]) and np.array_equal(sub[1], D2[:, 2699])


def test_core_distances_random_points_and_sweep_beyond_one_pass(eng):
    """Random points (no ties): the passes against numpy to rounding (numpy sums squares without fma), and the sweep
    itself with min_samples beyond one pass against the documented Prim rule."""
    from llckbdm_amd import _lib
    rng = np.random.default_rng(13)
    X = _data(rng, 2500, 3, 0.4, 200)
    D = np.sqrt(((X[:, None, :] - X[None, :, :]) ** 2).sum(-1))
    Ds = np.sort(D, axis=1)
    ks = [2560, 2690]
    got = eng.core_distances(X, ks)
    for f, k in enumerate(ks):
        np.testing.assert_allclose(got[f], Ds[:, k - 1], rtol=1e-14, atol=1e-14)
    labels, _ = eng.hdbscan_sweep(X, ks)
    lib = _lib.load()
    for f, k in enumerate(ks):
        e = _prim_reference(X, k)
        a = np.array([x[0] for x in e], dtype=np.int32)
        b = np.array([x[1] for x in e], dtype=np.int32)
        w = np.array([x[2] for x in e], dtype=np.float64)
        ref = np.empty(len(X), dtype=np.int32)
        lib.kbdm_hdbscan_labels_from_mst(len(X), _lib.ptr(a), _lib.ptr(b), _lib.ptr(w), 5, _lib.ptr(ref))
        assert np.array_equal(labels[f], ref), k


def test_silhouette_sweep_equals_one_call_per_labeling(eng):
    """The batched silhouettes (C ABI kbdm_silhouette_sweep, what llc_kbdm scores the sweep with) are the SAME BITS as one
    kbdm_silhouette_samples call per labeling, in the samples' original order; a labeling outside sklearn's precondition
    (one label value only) comes back marked invalid instead of raising."""
    rng = np.random.default_rng(17)
    X = _data(rng, 1200, 6, 0.3, 150)
    labels, _ = eng.hdbscan_sweep(X, [1, 3, 8, 20, 60])
    labs = np.concatenate([labels, np.zeros((1, len(X)), np.int32), rng.integers(-1, 9, (1, len(X))).astype(np.int32)])
    sil, ok = eng.silhouette_sweep(X, labs)
    assert sil.shape == labs.shape and ok.tolist() == [True] * 5 + [False, True]
    for f in range(len(labs)):
        if ok[f]:
            assert np.array_equal(sil[f], eng.silhouette_samples(X, labs[f])), f
    from sklearn.metrics import silhouette_samples
    np.testing.assert_allclose(sil[6], silhouette_samples(X, labs[6]), rtol=1e-10, atol=1e-12)
    eng.SWEEP_CHUNK_ENTRIES = 2 * len(X) + 5                       # two fits per call: the chunked path
    try:
        sil2, ok2 = eng.silhouette_sweep(X, labs)
    finally:
        del eng.SWEEP_CHUNK_ENTRIES
    assert np.array_equal(sil2, sil) and np.array_equal(ok2, ok)


def test_register_resident_prim_equals_the_general_kernel(eng):
    """k_prim_mst_reg (4-dimensional samples: a fit's state in registers / LDS; five size classes, the two largest with the
    core distances streamed) against k_prim_mst: the same points with a fifth coordinate of zeros take the general kernel
    and add exact zeros to every squared distance - the labels must be equal arrays, ties included (half of the points
    sit on a lattice)."""
    rng = np.random.default_rng(23)
    for n in (3000, 9000, 15000, 22000, 33000):
        X = _data(rng, n - n // 2, 8, 0.3, 0)
        lat = rng.integers(0, 12, (n // 2, 4)).astype(np.float64) * 0.25
        X = np.concatenate([X, lat])[rng.permutation(n)]
        ks = [1, 4, 25]
        a, na = eng.hdbscan_sweep(X, ks)
        b, nb = eng.hdbscan_sweep(np.column_stack([X, np.zeros(n)]), ks)
        assert np.array_equal(a, b), n
        assert np.array_equal(na, nb)


def test_llc_kbdm_with_batched_silhouettes_equals_fit_by_fit(eng):
    """llc_kbdm scores the sweep with ONE kbdm_silhouette_sweep call; an engine without that entry (the CPU tests' stand-in, an
    older library) is scored fit by fit: the same result, bit for bit."""
    from llckbdm_amd import datasets
    from llckbdm_amd.llckbdm import llc_kbdm
    sig = np.atleast_2d(datasets.config2(seed=3)[0])[0][:768]

    class FitByFit:
        def __getattr__(self, name):
            if name == "silhouette_sweep":
                raise AttributeError(name)
            return getattr(eng, name)
    a = llc_kbdm(sig, datasets.DWELL, range(80, 96), p=1, l=30, engine=eng)
    b = llc_kbdm(sig, datasets.DWELL, range(80, 96), p=1, l=30, engine=FitByFit())
    assert len(a.line_list) > 0
    assert np.array_equal(a.line_list, b.line_list) and a.rmse == b.rmse and np.array_equal(a.silhouette, b.silhouette)
