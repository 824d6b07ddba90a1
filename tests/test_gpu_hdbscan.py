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
