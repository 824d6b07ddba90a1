"""GPU (-m gpu): the rows next to the hot path through the C ABI - RMSE scoring kernel against
the vectors the reference produced, silhouette kernel against the direct restatement and scikit-learn,
min_rmse_kbdm / llc_kbdm / iterative_llc_kbdm as the reference's own tests exercise them."""
import os

import numpy as np
import pytest

from oracle import llc_oracle as L

pytestmark = pytest.mark.gpu

DWELL = 5e-4
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def eng():
    from llckbdm_amd.engine import Engine
    return Engine(0)


@pytest.fixture(scope="module")
def gnext():
    return np.load(os.path.join(HERE, "golden", "next_golden.npz"))


def test_rmse_kernel_matches_reference_vectors(eng, gnext):
    n = int(gnext["rmse_ncand"][0])
    cands = [gnext[f"rmse_cand{i}"] for i in range(n)]
    for name, data in (("clean", gnext["sig2048"]), ("noisy", gnext["noisy2048"])):
        got = eng.rmse_batch(data, DWELL, cands + [np.zeros((0, 4))])
        assert got[-1] == np.inf                                       # empty candidate (min_rmse_kbdm.py:36-37)
        for i in range(n):
            assert got[i] == pytest.approx(gnext[f"rmse_{name}{i}"][0], rel=1e-10, abs=1e-12)
    assert eng.rmse_batch(gnext["sig1000"], DWELL, [gnext["rmse_cand3"]])[0] == pytest.approx(gnext["rmse_odd"][0], rel=1e-10)


def test_calculate_freq_domain_rmse_like_the_reference_test(eng, gnext):
    """_tests/test_metrics.py:8-23."""
    from llckbdm_amd.metrics import calculate_freq_domain_rmse
    sig, params = gnext["sig2048"], gnext["params"]
    N = len(sig)
    assert calculate_freq_domain_rmse(data=sig, params_est=params, dwell=DWELL, engine=eng) == pytest.approx(0, abs=1e-12)
    rng = np.random.default_rng(0)
    noisy = sig + rng.standard_normal(N) + 1j * rng.standard_normal(N)
    want = np.sqrt(np.mean(((np.fft.fft(noisy) - np.fft.fft(sig)).real / np.sqrt(N)) ** 2))
    assert calculate_freq_domain_rmse(data=noisy, params_est=params, dwell=DWELL, engine=eng) == pytest.approx(want, rel=1e-10)
    with pytest.raises(ValueError):
        calculate_freq_domain_rmse(data=sig, params_est=[[1.0, -0.1, 10.0, 0.0]], dwell=DWELL, engine=eng)


def test_silhouette_kernel(eng):
    from sklearn.metrics import silhouette_samples
    labels = []
    rng = np.random.default_rng(5)
    centres = rng.standard_normal((12, 4))
    parts = []
    for i, c in enumerate(centres):
        k = int(rng.integers(2, 400))
        parts.append(rng.standard_normal((k, 4)) * 10.0 ** rng.uniform(-6, -1) + c)
        labels += [i] * k
    parts.append(rng.standard_normal((257, 4)))
    labels += [-1] * 256 + [12]                                         # noise class + one singleton
    X = np.concatenate(parts)
    labels = np.array(labels)
    perm = rng.permutation(len(X))
    X, labels = X[perm], labels[perm]
    got = eng.silhouette_samples(X, labels)
    assert np.abs(got - L.silhouette_samples_direct(X, labels)).max() < 1e-12
    assert np.abs(got - silhouette_samples(X, labels)).max() < 1e-6      # sklearn expands |x-y|^2: less exact on tight clusters
    assert got[labels == 12][0] == 0.0
    with pytest.raises(Exception):
        eng.silhouette_samples(X, np.zeros(len(X), dtype=int))          # one class only: undefined, as in sklearn


def test_min_rmse_kbdm_like_the_reference_test(eng, gnext):
    """_tests/test_min_rmse_kbdm.py:6-23."""
    from llckbdm_amd.min_rmse_kbdm import min_rmse_kbdm
    sig = gnext["sig2048"]
    m_range = [int(m) for m in gnext["minrmse_m_range"]]
    res = min_rmse_kbdm(data=sig, dwell=DWELL, m_range=m_range, l=30, engine=eng)
    assert len(res.samples) == len(m_range)
    assert res.min_rmse == pytest.approx(0, abs=1e-10)
    assert res.min_index == 2 == int(gnext["minrmse_index"][0])
    # the ill-posed members (m ~ 30 < 2 x 16 peaks) score like the reference's: same order of magnitude
    ref = gnext["minrmse_rmses"]
    for i in (0, 1, 3, 4, 5):
        assert 0.2 * ref[i] < res.rmses_list[i] < 5 * ref[i]
    # a noisy ensemble: the scores of the GPU line lists equal the oracle's scores of the same line lists
    noisy = gnext["noisy2048"]
    res = min_rmse_kbdm(data=noisy, dwell=DWELL, m_range=range(100, 110), engine=eng)
    want = [L.calculate_freq_domain_rmse(noisy, s, DWELL) for s in res.samples]
    assert res.rmses_list == pytest.approx(want, rel=1e-9)
    assert res.rmses_list == pytest.approx(list(gnext["minrmse2_rmses"]), rel=1e-6)     # and the reference's own run
    assert res.min_index == int(gnext["minrmse2_index"][0])


def test_llc_kbdm_like_the_reference_test(eng, gnext):
    """_tests/test_llckbdm.py:40-69."""
    from llckbdm_amd.llckbdm import llc_kbdm
    from llckbdm_amd.sampling import filter_samples
    from llckbdm_amd.sig_gen import multi_fid
    sig, params = gnext["sig2048"], gnext["params"]
    with pytest.raises(ValueError) as e:
        llc_kbdm(data=sig, dwell=DWELL, m_range=[1], engine=eng)
    assert "size of 'm_range' must be greater than 2" in str(e.value)
    results = llc_kbdm(data=sig, dwell=DWELL, m_range=range(250, 260, 1), p=1, l=30, engine=eng)
    line_list = filter_samples(results.line_list, amplitude_tol=1e-3)
    assert len(line_list) == len(params)
    est = multi_fid(np.arange(len(sig)) * DWELL, line_list)
    assert np.std(est.real - sig.real) < 1e-3 and np.std(est.imag - sig.imag) < 1e-3
    # parity of the consumer as SURVEY.md 8c defines it: the same clusterer on GPU-produced and on
    # CPU(oracle)-produced line lists gives the same partition sizes and the same summarised lines
    from tests.fake_engine import OracleEngine
    ref = llc_kbdm(data=sig, dwell=DWELL, m_range=range(250, 260, 1), p=1, l=30, engine=OracleEngine())
    a = line_list[np.argsort(line_list[:, 2])]
    b = filter_samples(ref.line_list, amplitude_tol=1e-3)
    b = b[np.argsort(b[:, 2])]
    assert a.shape == b.shape and np.abs(a[:, :3] / b[:, :3] - 1).max() < 1e-6
    assert results.rmse == pytest.approx(ref.rmse, abs=1e-9)


def test_iterative_llc_kbdm_runs(eng, gnext):
    """_tests/test_llckbdm.py:82-87 (a smoke test in the reference too)."""
    from llckbdm_amd.llckbdm import iterative_llc_kbdm
    res = iterative_llc_kbdm(data=gnext["sig2048"], dwell=DWELL, m_range=range(180, 190), engine=eng, max_iterations=2)
    # every pass drops the lines at or below the silhouette percentile (llckbdm.py:177-181): 16 peaks -> 15 kept
    assert 12 <= len(res.line_list) <= 32 and res.rmse is not None and res.rmse < 1e-6
    assert len(res.line_lists) == len(res.silhouettes) >= 1
    with pytest.raises(ValueError):
        iterative_llc_kbdm(data=gnext["sig2048"], dwell=DWELL, m_range=range(180, 190), max_iterations=0, engine=eng)
