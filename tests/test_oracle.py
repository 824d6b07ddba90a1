"""CPU: pin the oracle (our numpy/scipy restatement) against vectors made by the reference."""
import numpy as np
import pytest

from oracle import kbdm_oracle as O
from tests.helpers import canonical, assert_lines_close, genuine_rows


ILL_POSED = ("m30", "m10q")


def _case(golden, name):
    m, l, p = (int(x) for x in golden[f"{name}__meta"])
    q = float(golden[f"{name}__q"][0])
    sig = golden[str(golden[f"{name}__sig"])]
    return sig, m, l, p, q


def test_signal_matches_reference_fixture(golden):
    # reference _tests/fixtures.py:44-47 builds the signal with sig_gen.multi_fid
    assert np.array_equal(O.brain_sim_params_sorted(), golden["params_sorted"])
    assert np.array_equal(O.brain_sim_signal(2048), golden["sig2048"])
    assert np.array_equal(O.brain_sim_signal(1024), golden["sig1024"])
    assert np.array_equal(O.make_noisy(golden["sig2048"], 1e-3, 0), golden["sig2048_n3"])
    assert np.array_equal(O.make_noisy(golden["sig2048"], 1e-6, 7), golden["sig2048_n6"])


def test_hankel_bit_exact(golden):
    sig = golden["sig2048"]
    U0, Up1, Up = O.compute_U_matrices(sig, 300, 2)
    assert np.array_equal(np.stack([U0[0], U0[-1]]), golden["hankel_p2_m300_U0_rows"])
    assert np.array_equal(np.stack([Up1[0], Up1[-1]]), golden["hankel_p2_m300_Up1_rows"])
    assert np.array_equal(np.stack([Up[0], Up[-1]]), golden["hankel_p2_m300_Up_rows"])
    U0, Up1, Up = O.compute_U_matrices(sig, 17, 3)
    assert np.array_equal(U0, golden["hankel_p3_m17_U0"])
    assert np.array_equal(Up1, golden["hankel_p3_m17_Up1"])
    assert np.array_equal(Up, golden["hankel_p3_m17_Up"])


@pytest.mark.parametrize("name", ["c1", "m300", "m150", "m100", "m101", "m102", "m30", "m10q",
                                  "m180l30", "m64p2", "n3m128", "n3m256", "n6m256", "n3m512"])
def test_kbdm_matches_reference(golden, name):
    sig, m, l, p, q = _case(golden, name)
    ll, info = O.kbdm(sig, 5e-4, m=m, p=p, l=(None if l == m else l), q=q)
    assert ll.shape == (l, 4)
    assert (info.m, info.l, info.p) == (m, l, p)
    sv = golden[f"{name}__sv"]
    assert info.singular_values.shape == sv.shape
    # singular values carry absolute error ~eps*s0 (backward stability of the SVD)
    assert np.max(np.abs(info.singular_values - sv)) < 1e-13 * sv[0]
    kept = canonical(O.filter_samples(ll))
    want = golden[f"{name}__kept"]
    if name in ILL_POSED:
        # rank-deficient by construction (m < 2 x peaks, or q>0 at m=10): spurious lines sit at
        # the 1e-6 filter threshold and move with BLAS kernel/threading; pin the strong lines
        kept, want = kept[kept[:, 0] > 1e-2], want[want[:, 0] > 1e-2]
        assert len(kept) == len(want)
        assert_lines_close(kept, want, rel=1e-5, phase_abs=1e-5, what=name)
        return
    assert len(kept) == len(want)
    # same LAPACK, same numpy, same inputs, same operation order; BLAS thread count and CPU
    # kernel selection may differ from the generating run, hence not bit-exact
    tol = 1e-6 if name == "n6m256" else 1e-8
    assert_lines_close(kept, want, rel=tol, phase_abs=tol, what=name)
    if name in ("c1", "m300", "m150", "m100"):
        truth = golden["params_sorted"]
        assert_lines_close(genuine_rows(kept, truth), genuine_rows(want, truth), rel=1e-9,
                           phase_abs=1e-9, what=name + " genuine")


def test_validation_errors(golden):
    sig = golden["sig2048"]
    with pytest.raises(ValueError, match="l or m must be specified"):
        O.kbdm(sig, 5e-4)
    with pytest.raises(ValueError, match="l can't be greater than m"):
        O.kbdm(sig, 5e-4, l=30, m=20)
    with pytest.raises(ValueError, match=r"m or l can't be greater than \(n \+ 1 - p\)/2\."):
        O.kbdm(sig, 5e-4, m=len(sig) // 2 + 1)


def test_sampler_matches_reference(golden):
    lls, infos = O.sample_kbdm(golden["sig2048"], 5e-4, range(100, 103), p=1, l=None, q=0)
    assert [len(x) for x in lls] == list(golden["sample_100_103_counts"])
    assert [i.m for i in infos] == list(golden["sample_100_103_ms"])
    for i, x in enumerate(lls):
        assert_lines_close(canonical(x), golden[f"sample_100_103_ll{i}"], rel=1e-7, phase_abs=1e-7)


def test_filter_empty():
    e = np.array([])
    assert np.array_equal(O.filter_samples(e), e)


@pytest.mark.parametrize("name", ["n3m256q", "c4m611"])
def test_kbdm_matches_reference_round2_goldens(name):
    """Round-2 vectors (tests/golden/make_golden_r2.py, made by the reference itself): a well-posed Tikhonov case
    (kbdm.py:179-184) and a config-4 member (N = 4096, 32 peaks, m = 611)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with np.load(os.path.join(root, "tests", "golden", "kbdm_golden_r2.npz")) as z:
        g = {k: z[k] for k in z.files}
    sig, m, l, p, q = _case(g, name)
    ll, info = O.kbdm(sig, 5e-4, m=m, p=p, l=(None if l == m else l), q=q)
    sv = g[f"{name}__sv"]
    assert np.max(np.abs(info.singular_values - sv)) < 1e-13 * sv[0]
    kept, want = canonical(O.filter_samples(ll)), g[f"{name}__kept"]
    assert len(kept) == len(want)
    assert_lines_close(kept, want, rel=1e-8, phase_abs=1e-8, what=name)
    if name == "c4m611":
        # the signal the product's datasets module builds for config 4 is the one the reference's sig_gen gave
        from llckbdm_amd import datasets
        assert np.allclose(datasets.config4()[0][0], g["sig4096_c4"], rtol=0, atol=1e-13)
