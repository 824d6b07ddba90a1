"""CPU: host-side logic that mirrors the reference's API (no GPU compute involved)."""
import numpy as np
import pytest

from llckbdm_amd import sig_gen
from llckbdm_amd.kbdm import _resolve_m_l, KbdmInfo
from llckbdm_amd.sampling import filter_samples
from oracle import kbdm_oracle as O


def test_argument_checks_match_reference_strings():
    # reference test_kbdm.py:62-103
    with pytest.raises(ValueError, match="l or m must be specified"):
        _resolve_m_l(2048, None, 1, None)
    with pytest.raises(ValueError, match="l can't be greater than m"):
        _resolve_m_l(2048, 20, 1, 30)
    assert _resolve_m_l(2048, None, 1, 30) == (30, 30)
    assert _resolve_m_l(2048, 30, 1, None) == (30, 30)
    for m, l in ((1025, None), (1025, 1025), (1025, 10)):
        with pytest.raises(ValueError, match=r"m or l can't be greater than \(n \+ 1 - p\)/2\."):
            _resolve_m_l(2048, m, 1, l)
    assert _resolve_m_l(2048, 1024, 1, None) == (1024, 1024)
    with pytest.raises(ValueError):
        _resolve_m_l(2048, 1024, 2, None)


def test_kbdm_raises_before_touching_the_gpu():
    from llckbdm_amd.kbdm import kbdm
    with pytest.raises(ValueError, match="l or m must be specified"):
        kbdm(np.ones(64, dtype=complex), 1e-3)


def test_kbdm_info_fields():
    info = KbdmInfo(m=3, l=2, p=1, q=0, singular_values=np.arange(3.0))
    assert (info.m, info.l, info.p, info.q) == (3, 2, 1, 0)


def test_filter_samples_matches_oracle():
    rng = np.random.default_rng(0)
    ll = rng.standard_normal((50, 4))
    ll[:, 0] = np.abs(ll[:, 0]) * 1e-5
    assert np.array_equal(filter_samples(ll), O.filter_samples(ll))
    e = np.array([])
    assert np.array_equal(filter_samples(e), e)       # reference test_sampling.py:65-68


def test_sig_gen_matches_oracle(golden):
    t = np.linspace(0, 5e-4 * 2048, 2048, endpoint=False)
    assert np.array_equal(sig_gen.multi_fid(t, golden["params_sorted"]), golden["sig2048"])
    t2, f = sig_gen.gen_t_freq_arrays(1000, 5e-4)
    assert len(t2) == 1000 and len(f) == 1000
    with pytest.raises(ValueError, match="T2 must be positive."):
        sig_gen.fid(t, 1.0, 0.0, 1.0)
    with pytest.raises(ValueError, match="Amplitude can't be negative."):
        sig_gen.fid(t, -1.0, 1.0, 1.0)
    # lorentzian of a peak == analytic FT of its FID (reference test_sig_gen checks the same property)
    p = (1.0, 0.05, 100.0, 0.3)
    assert sig_gen.lorentzian_peak(np.array([100.0]), *p)[0] == pytest.approx(0.05 * np.exp(0.3j))


def test_non_finite_samples_raise_valueerror_as_scipy_does():
    """scipy.linalg.svd / eig run with check_finite=True in the reference (kbdm.py:166,192): a NaN / Inf among the samples a
    member uses raises ValueError("array must not contain infs or NaNs"); one beyond its window goes unnoticed.  Checked on
    the host, before any GPU work (a fake engine records that nothing was submitted)."""
    import numpy as np
    import pytest
    from llckbdm_amd.kbdm import kbdm
    from llckbdm_amd.sampling import sample_kbdm, sample_kbdm_signals
    from tests.fake_engine import OracleEngine

    class FakeEngine(OracleEngine):
        calls = 0

        def solve(self, *a, **k):
            self.calls += 1
            return super().solve(*a, **k)
    sig = np.exp((-0.01 + 0.3j) * np.arange(256)) + 1e-3 * np.random.default_rng(0).standard_normal(256)
    for idx, val in ((5, np.nan), (30, np.inf)):
        bad = sig.copy(); bad[idx] = val
        eng = FakeEngine()
        with pytest.raises(ValueError, match="array must not contain infs or NaNs"):
            kbdm(bad, 5e-4, m=16, p=1, engine=eng)
        with pytest.raises(ValueError, match="array must not contain infs or NaNs"):
            sample_kbdm(bad, 5e-4, range(8, 17, 2), p=1, l=None, engine=eng)
        with pytest.raises(ValueError, match="array must not contain infs or NaNs"):
            sample_kbdm_signals(np.stack([sig, bad]), 5e-4, [0, 1, 1], [16, 12, 16], engine=eng)
        assert eng.calls == 0
    far = sig.copy(); far[200] = np.nan                       # outside data[0 : 2 m + p - 1] = data[0:32]
    eng = FakeEngine()
    kbdm(far, 5e-4, m=16, p=1, engine=eng)
    assert eng.calls == 1
