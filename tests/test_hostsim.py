"""CPU: the algorithm templates the HIP kernels instantiate (llckbdm_amd/csrc/kb_*.hpp), run
through the one-thread host context of tests/hostsim, reproduce the reference's golden vectors.
This checks index/convergence logic without a GPU; the GPU parity tests check the real kernels."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from tests.helpers import canonical, assert_lines_close, genuine_rows, keep_mask

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = ctypes.c_void_p


@pytest.fixture(scope="module")
def hs():
    src = os.path.join(ROOT, "tests", "hostsim", "hostsim.cpp")
    out_dir = os.path.join(ROOT, "tests", "hostsim", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libhostsim.so")
    inc = os.path.join(ROOT, "llckbdm_amd", "csrc")
    deps = [src] + [os.path.join(inc, f) for f in os.listdir(inc)]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-shared", "-fPIC", "-I", inc, src, "-o", so], check=True)
    return ctypes.CDLL(so)


def _kbdm(hs, sig, m, l, p, q, dwell=5e-4):
    sig = np.ascontiguousarray(sig, dtype=complex)
    lines, sv, mu = np.zeros((l, 4)), np.zeros(m), np.zeros(l, complex)
    info = hs.hs_kbdm(sig.ctypes.data_as(P), len(sig), m, l, p, ctypes.c_double(q), ctypes.c_double(dwell),
                      lines.ctypes.data_as(P), sv.ctypes.data_as(P), mu.ctypes.data_as(P))
    return lines, sv, mu, info


def test_svd_random_and_degenerate(hs):
    rng = np.random.default_rng(1)
    mats = [rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m)) for m in (1, 2, 7, 33)]
    mats.append(np.zeros((5, 5), complex))
    mats.append(np.eye(6, dtype=complex))
    mats.append(np.diag([3.0, -2.0, 0.0, 1e-20]).astype(complex))
    for A in mats:
        m = A.shape[0]
        Af = np.asfortranarray(A)
        L, R, s = np.zeros((m, m), complex, order="F"), np.zeros((m, m), complex, order="F"), np.zeros(m)
        info = hs.hs_svd(Af.ctypes.data_as(P), m, L.ctypes.data_as(P), s.ctypes.data_as(P), R.ctypes.data_as(P))
        assert info == 0
        scale = max(1.0, np.abs(A).max())
        assert np.abs(L @ np.diag(s) @ R.conj().T - A).max() < 1e-13 * scale * m
        assert np.abs(L.conj().T @ L - np.eye(m)).max() < 1e-13 * m
        assert np.abs(R.conj().T @ R - np.eye(m)).max() < 1e-13 * m
        assert np.all(np.diff(s) <= 0) and np.all(s >= 0)
        assert np.abs(s - np.linalg.svd(A, compute_uv=False)).max() < 1e-13 * scale * m


def test_householder_generator_survives_underflow_and_overflow_of_the_squares(hs):
    """zlarfg rescales when the squares of a vector's entries underflow or overflow; so does `larfg`: matrices scaled by
    1e-200 / 1e+200 (bidiagonalisation) and a Hessenberg matrix with subdiagonals between 1e-160 and denormal (the
    Hessenberg reduction in front of the QR iteration; found as NaN eigenvalues of a member with a 1e-300 subdiagonal)."""
    rng = np.random.default_rng(12)
    for scale in (1e-200, 1e200):
        m = 20
        A = (rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m))) * scale
        Af = np.asfortranarray(A)
        L, R, s = np.zeros((m, m), complex, order="F"), np.zeros((m, m), complex, order="F"), np.zeros(m)
        assert hs.hs_svd(Af.ctypes.data_as(P), m, L.ctypes.data_as(P), s.ctypes.data_as(P), R.ctypes.data_as(P)) == 0
        ref = np.linalg.svd(A / scale, compute_uv=False)
        assert np.isfinite(s).all() and np.abs(s / scale - ref).max() < 1e-13 * m * ref[0]
        assert np.abs((L * (s / scale)) @ R.conj().T - A / scale).max() < 1e-12 * m
    hs.hs_eigvals2.argtypes = [P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, P, ctypes.c_int]
    for val in (1e-160, 1e-200, 1e-300, 1e-310):
        n = 60
        W = np.triu(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)), -1)
        W[25, 24] = val
        Wf = np.asfortranarray(W)
        mu, st = np.zeros(n, complex), np.zeros(16, np.int64)
        assert hs.hs_eigvals2(Wf.ctypes.data_as(P), n, 8, 56, 0, mu.ctypes.data_as(P), st.ctypes.data_as(P), 0) == 0
        ref = np.linalg.eigvals(W)
        assert np.abs(mu[:, None] - ref[None, :]).min(axis=1).max() < 1e-11 * n


def test_eig_random(hs):
    rng = np.random.default_rng(2)
    for n in (1, 2, 3, 16, 40, 96, 130):     # 96 and 130: one and two blocked Hessenberg panels (+ k_hess_z's reference form)
        W = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        Wf = np.asfortranarray(W)
        mu, Pm = np.zeros(n, complex), np.zeros((n, n), complex, order="F")
        info = hs.hs_eig(Wf.ctypes.data_as(P), n, mu.ctypes.data_as(P), Pm.ctypes.data_as(P))
        assert info == 0
        assert np.abs(W @ Pm - Pm * mu[None, :]).max() < 1e-12 * n
        ref = np.linalg.eigvals(W)
        assert np.abs(mu[:, None] - ref[None, :]).min(axis=1).max() < 1e-11


def test_aberth_shift_solver(hs):
    """Ehrlich-Aberth on Hyman's recurrence (kb_hqr_ms.hpp::aberth_eigs): all eigenvalues of small unreduced
    Hessenberg matrices to a few ulp - random ones and the nearly triangular trailing blocks of a QR iteration
    in progress (tiny subdiagonals); and the multishift iteration that uses it hardly ever falls back."""
    import scipy.linalg as sla
    rng = np.random.default_rng(11)
    for n in (3, 5, 8, 16, 24):
        for trial in range(6):
            A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
            T = sla.hessenberg(A)
            if trial >= 3:      # graded subdiagonal, as in a converging trailing block
                for k in range(1, n):
                    T[k, k - 1] *= 10.0 ** (-2.0 * trial * k / n)
            Tf = np.asfortranarray(T)
            z = np.zeros(n, complex)
            ok = hs.hs_aberth(Tf.ctypes.data_as(P), n, z.ctypes.data_as(P))
            assert ok == 1, (n, trial)
            ref = np.linalg.eigvals(T)
            err = np.abs(z[:, None] - ref[None, :]).min(axis=1).max()
            assert err < 1e-12 * np.abs(T).max(), (n, trial, err)
            # every reference eigenvalue is found exactly once
            assert len(set(np.abs(z[:, None] - ref[None, :]).argmin(axis=1))) == n
    # exactly reducible matrix: the solver declines (caller falls back to the QR solver)
    T = np.asfortranarray(np.triu(rng.standard_normal((6, 6)) + 0j))
    z = np.zeros(6, complex)
    assert hs.hs_aberth(T.ctypes.data_as(P), 6, z.ctypes.data_as(P)) == 0
    # inside the multishift iteration
    n = 150
    W = np.asfortranarray(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    mu, st = np.zeros(n, complex), np.zeros(16, np.int64)
    hs.hs_eigvals2.argtypes = [P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, P, ctypes.c_int]
    assert hs.hs_eigvals2(W.ctypes.data_as(P), n, 8, 56, 0, mu.ctypes.data_as(P), st.ctypes.data_as(P), 0) == 0
    ref = np.linalg.eigvals(W)
    assert np.abs(mu[:, None] - ref[None, :]).min(axis=1).max() < 1e-11
    assert st[8] > 10 and st[9] <= st[8] // 10, st          # Aberth calls / fallbacks to the small QR iteration


@pytest.mark.parametrize("name", ["m100", "m64p2", "m180l30", "m10q", "n3m128"])
def test_pipeline_matches_reference_golden(hs, golden, name):
    m, l, p = (int(x) for x in golden[f"{name}__meta"])
    q = float(golden[f"{name}__q"][0])
    sig = golden[str(golden[f"{name}__sig"])]
    ll, sv, mu, info = _kbdm(hs, sig, m, l, p, q)
    assert info == 0
    ref_sv = golden[f"{name}__sv"]
    assert np.abs(sv - ref_sv).max() < 1e-13 * ref_sv[0] * m
    kept = canonical(ll[keep_mask(ll)])
    want = golden[f"{name}__kept"]
    assert len(kept) == len(want)
    assert_lines_close(kept, want, rel=1e-7, phase_abs=1e-7, what=name)
    if name == "m100":
        truth = golden["params_sorted"]
        assert_lines_close(genuine_rows(kept, truth), genuine_rows(want, truth), rel=1e-8, phase_abs=1e-8)


def test_second_generation_qr_iteration_host_form(hs):
    """kb_hqr2.hpp on the host context: double-shift bulges (3-element reflectors, 2 nb shifts per sweep), the
    time-major log, strip units and the team split (helper's share inline).  Eigenvalues against LAPACK, solo and team
    bit-identical, protocol counters consistent; also on a reduced KBDM matrix (the shape the kernel sees)."""
    rng = np.random.default_rng(9)
    hs.hs_eigvals2.argtypes = [P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, P, ctypes.c_int]

    def run2(W, nb, win, team):
        n = W.shape[0]
        Wf = np.asfortranarray(W)
        mu, st = np.zeros(n, complex), np.zeros(16, np.int64)
        info = hs.hs_eigvals2(Wf.ctypes.data_as(P), n, nb, win, team, mu.ctypes.data_as(P), st.ctypes.data_as(P), 0)
        return info, mu, st

    for n, nb, win in ((3, 8, 56), (9, 8, 56), (13, 8, 56), (40, 4, 32), (70, 8, 56), (130, 8, 56), (90, 2, 24)):
        W = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        i_s, mu_s, st_s = run2(W, nb, win, 0)
        i_t, mu_t, st_t = run2(W, nb, win, 1)
        assert i_s == 0 and i_t == 0
        assert np.array_equal(mu_s, mu_t)
        ref = np.linalg.eigvals(W)
        assert np.abs(mu_s[:, None] - ref[None, :]).min(axis=1).max() < 1e-11 * n
        assert len(set(np.abs(mu_s[:, None] - ref[None, :]).argmin(axis=1))) == n          # every eigenvalue exactly once
        published, all_done, done, near_done = (int(x) for x in st_t[4:8])
        assert published == st_t[3] and all_done == published and near_done == published and done == 1
    # a reduced KBDM matrix (the shape the kernel sees): Hankel signal -> SVD -> projected U^1
    import scipy.linalg as sla
    from oracle import kbdm_oracle as O
    sig = O.make_noisy(O.brain_sim_signal(1024), 1e-3, 0)
    m = 150
    U0, Up1, Up = O.compute_U_matrices(sig, m, 1)
    L, s, Rh = sla.svd(Up1)
    D = np.diag(1 / np.sqrt(s))
    W = D @ L.conj().T @ Up @ Rh.conj().T @ D
    info, mu, st = run2(W, 8, 56, 0)
    assert info == 0
    ref = np.linalg.eigvals(W)
    assert np.abs(mu[:, None] - ref[None, :]).min(axis=1).max() < 1e-11 * np.abs(ref).max() * m


def _bdsdc(hs, d, e):
    m = len(d)
    hs.hs_bdsdc.argtypes = [P, P, ctypes.c_int, P, P, P]
    d = np.ascontiguousarray(d, float)
    e = np.ascontiguousarray(np.r_[e, 0.0], float)
    X, Y, s = np.zeros((m, m), order="F"), np.zeros((m, m), order="F"), np.zeros(m)
    info = hs.hs_bdsdc(d.ctypes.data_as(P), e.ctypes.data_as(P), m, X.ctypes.data_as(P), s.ctypes.data_as(P), Y.ctypes.data_as(P))
    return X, s, Y, info


def _check_bdsdc(hs, d, e, tol=2e-14):
    n = len(d)
    B = np.diag(d) + (np.diag(e, 1) if n > 1 else 0)
    X, s, Y, info = _bdsdc(hs, d, e)
    sref = np.linalg.svd(B, compute_uv=False)
    s0 = max(sref[0], 1e-300)
    assert info == 0
    assert np.all(np.diff(s) <= 0) and np.all(s >= 0)
    assert np.abs(s - sref).max() <= tol * s0
    assert np.abs(X * s @ Y.T - B).max() <= tol * s0
    assert np.abs(X.T @ X - np.eye(n)).max() <= tol and np.abs(Y.T @ Y - np.eye(n)).max() <= 2 * tol


def test_bidiagonal_divide_and_conquer(hs):
    """kb_bdsdc.hpp (the templates k_dc_* instantiate) against numpy: random, graded, clustered, rank-deficient,
    badly scaled and all-zero bidiagonals, sizes across the leaf / tree boundaries."""
    rng = np.random.default_rng(7)
    for n in (1, 2, 5, 32, 33, 65, 66, 130, 257):
        _check_bdsdc(hs, rng.standard_normal(n), rng.standard_normal(n - 1))
    _check_bdsdc(hs, np.ones(64), np.ones(63))
    _check_bdsdc(hs, 2.0 ** -np.arange(60), 2.0 ** -np.arange(59))
    _check_bdsdc(hs, np.r_[np.ones(30), 1e-9 * np.ones(30)], 1e-12 * np.ones(59))
    _check_bdsdc(hs, np.zeros(40), np.ones(39))
    _check_bdsdc(hs, np.ones(40), np.zeros(39))
    _check_bdsdc(hs, np.zeros(70), np.zeros(69))
    for trial in range(40):
        n = int(rng.integers(1, 200))
        d, e = rng.standard_normal(n), rng.standard_normal(max(n - 1, 0))
        kind = trial % 8
        if kind == 1: d *= 10.0 ** rng.uniform(-12, 0, n)
        if kind == 2: e *= 10.0 ** rng.uniform(-16, 0, max(n - 1, 0))
        if kind == 3: d[rng.random(n) < 0.2] = 0.0
        if kind == 4: e[rng.random(max(n - 1, 0)) < 0.2] = 0.0
        if kind == 5: d, e = np.round(d * 2) / 2, np.round(e * 2) / 2
        if kind == 6: d, e = d * 1e150, e * 1e150
        if kind == 7: d, e = d * 1e-150, e * 1e-150
        _check_bdsdc(hs, d, e)


def test_divide_and_conquer_on_kbdm_hankel_bidiagonals(hs):
    """The bidiagonal forms this path actually sees: Hankel matrices of the brain-sim signal, noise-free (singular values
    down to 1e-16 s0: heavy deflation) and noisy, through hs_bidiag + hs_bdsdc."""
    from oracle import kbdm_oracle as O
    import scipy.linalg as sl
    hs.hs_bidiag.argtypes = [P, ctypes.c_int, P, P, P, P]
    for sigma, m, N in ((0.0, 150, 1024), (1e-3, 200, 2048), (1e-6, 256, 2048)):
        sig = O.brain_sim_signal(N)
        if sigma:
            sig = O.make_noisy(sig, sigma, 1)
        A = np.asfortranarray(sl.hankel(sig[:m], sig[m - 1:2 * m - 1]))
        d, e = np.zeros(m), np.zeros(m)
        Q, Pm = np.zeros((m, m), complex, order="F"), np.zeros((m, m), complex, order="F")
        hs.hs_bidiag(A.ctypes.data_as(P), m, d.ctypes.data_as(P), e.ctypes.data_as(P), Q.ctypes.data_as(P), Pm.ctypes.data_as(P))
        _check_bdsdc(hs, d, e[:m - 1])
        X, s, Y, _ = _bdsdc(hs, d, e[:m - 1])
        s_ref = np.linalg.svd(A, compute_uv=False)
        assert np.abs(s - s_ref).max() <= 1e-14 * m * s_ref[0]
        L, R = Q @ X, Pm @ Y
        assert np.abs(L * s @ R.conj().T - A).max() <= 1e-13 * s_ref[0]


def test_divide_and_conquer_aberth_eigenvalues(hs):
    """kb_aberth.hpp (host reference of the device's fast eigenvalue path): random matrices and reduced KBDM matrices
    (noise-free, noisy, l < m, q > 0) - every eigenvalue of LAPACK exactly once, a few iterations per root and level;
    a matrix that splits is declined (the caller runs the QR iteration)."""
    import scipy.linalg as sla
    from oracle import kbdm_oracle as O
    hs.hs_eig_aberth.argtypes = [P, ctypes.c_int, P, P]
    rng = np.random.default_rng(4)

    def run(W):
        n = W.shape[0]
        Wf = np.asfortranarray(W, dtype=complex)
        mu, out = np.zeros(n, complex), np.zeros(2, np.int64)
        assert hs.hs_eig_aberth(Wf.ctypes.data_as(P), n, mu.ctypes.data_as(P), out.ctypes.data_as(P)) == 0
        return mu, int(out[0]), int(out[1])

    def reduced(sig, m, l=None, q=0.0):
        U0, Up1, Up = O.compute_U_matrices(sig, m, 1)
        L, s, Rh = sla.svd(Up1)
        l = l or m
        d = 1 / np.sqrt(s[:l] + (q * q / s[:l] if q else 0))
        return (d[:, None] * (L[:, :l].conj().T @ Up @ Rh[:l].conj().T)) * d[None, :]

    cases = [rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)) for n in (5, 33, 70, 150)]
    sig = O.make_noisy(O.brain_sim_signal(1024), 1e-3, 0)
    cases += [reduced(sig, 150), reduced(O.brain_sim_signal(1024), 120), reduced(sig, 140, l=60), reduced(sig, 130, q=1e-3)]
    for W in cases:
        n = W.shape[0]
        mu, declined, iters = run(W)
        assert declined == 0
        ref = np.linalg.eigvals(W)
        d = np.abs(mu[:, None] - ref[None, :])
        assert d.min(axis=1).max() < 1e-11 * max(1.0, np.abs(ref).max()) * n
        assert len(set(d.argmin(axis=1))) == n
        if n > 32:
            assert iters < 14 * n * np.log2(n / 16)          # a handful of iterations per root and level
    A = rng.standard_normal((80, 80)) + 0j
    A = np.triu(A, -1)
    A[40, 39] = 0.0                                            # an exact split
    assert run(A)[1] == 1


def test_cooperative_panels_give_the_same_bits_for_every_team_size(hs, monkeypatch):
    """kb_team.hpp / kb_panel_team.hpp: the panels of the two blocked reductions run by teams of T "workgroups" (here T
    threads that meet in a barrier, sharing the matrices and the exchange buffer as the device's workgroups do) return the
    same BITS for every T - the split products are defined by a fixed slot decomposition, T only deals the slots - and
    they are correct reductions (singular values / eigenvalues against LAPACK, reconstruction)."""
    rng = np.random.default_rng(5)

    def bidiag(A):
        m = A.shape[0]
        Af = np.asfortranarray(A)
        d, e = np.zeros(m), np.zeros(m)
        Q, Pm = np.zeros((m, m), complex, order="F"), np.zeros((m, m), complex, order="F")
        hs.hs_bidiag(Af.ctypes.data_as(P), m, d.ctypes.data_as(P), e.ctypes.data_as(P), Q.ctypes.data_as(P), Pm.ctypes.data_as(P))
        return d, e, Q, Pm

    def eig(W):
        n = W.shape[0]
        Wf = np.asfortranarray(W)
        mu, Pm = np.zeros(n, complex), np.zeros((n, n), complex, order="F")
        info = hs.hs_eig(Wf.ctypes.data_as(P), n, mu.ctypes.data_as(P), Pm.ctypes.data_as(P))
        return mu, Pm, info

    for m in (97, 161):                     # one and three panels of 32 columns
        A = rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m))
        W = rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m))
        sv0, mu0 = np.linalg.svd(A, compute_uv=False), np.linalg.eigvals(W)
        base = None
        for T in (1, 2, 3, 8):
            monkeypatch.setenv("HS_PANEL_T", str(T))
            d, e, Q, Pm = bidiag(A)
            mu, X, info = eig(W)
            assert info == 0
            if base is None:
                base = (d, e, Q, Pm, mu, X)
                B = np.diag(d) + np.diag(e[:m - 1], 1)
                assert np.abs(Q @ B @ Pm.conj().T - A).max() < 1e-13 * m
                assert np.abs(np.linalg.svd(B, compute_uv=False) - sv0).max() < 1e-12 * m
                assert np.abs(mu[:, None] - mu0[None, :]).min(axis=1).max() < 1e-10
                assert np.abs(W @ X - X * mu).max() < 1e-12 * m
            else:
                for a, b in zip(base, (d, e, Q, Pm, mu, X)):
                    assert np.array_equal(a, b), f"team of {T} differs from a team of one (m = {m})"
        monkeypatch.delenv("HS_PANEL_T")
