"""GPU: the acceptance of the divide-and-conquer Ehrlich-Aberth eigenvalue path (kb_aberth.hpp, k_ab_*; reference
kbdm.py:192, scipy.linalg.eig) on spectra that are hard for it: clusters, a defective repeated eigenvalue, the nearly
unitary reduced matrices of noise-free KBDM members at l = m = 512.  The checks are multiset matches with LAPACK in BOTH
directions (every computed root next to an eigenvalue AND every eigenvalue next to a computed root, one-to-one where the
spacing allows it): two approximations parked on one eigenvalue - the failure mode a trace / trace^2 check barely sees -
cannot pass.  A member the path does not trust must come back through the QR iteration (`last_eig_fallbacks`), never as a
wrong root."""
import numpy as np
import pytest
import scipy.linalg as sla

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from llckbdm_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _unitary(rng, n):
    q, r = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    return q * (np.diag(r) / np.abs(np.diag(r)))


def _one_to_one(mu, ref, tol):
    """Greedy one-to-one matching of two multisets of complex numbers; the largest pair distance must stay below tol."""
    mu, ref = list(mu), list(ref)
    assert len(mu) == len(ref)
    worst = 0.0
    for z in sorted(mu, key=lambda v: (v.real, v.imag)):
        k = int(np.argmin(np.abs(np.asarray(ref) - z)))
        worst = max(worst, abs(ref[k] - z))
        ref.pop(k)
    assert worst < tol, worst
    return worst


def test_cluster_of_fifty_eigenvalues_of_a_normal_matrix(eng):
    """50 eigenvalues spaced 1e-10 apart (plus 150 spread ones) of a NORMAL matrix: every eigenvalue is perfectly
    conditioned, so LAPACK resolves each of them, and so must this path - one root per eigenvalue, to 1e-13 ||A||."""
    rng = np.random.default_rng(21)
    n = 200
    lam = np.concatenate([0.7 + 0.2j + 1e-10 * np.arange(50), np.exp(2j * np.pi * rng.random(150)) * (0.3 + 0.7 * rng.random(150))])
    Q = _unitary(rng, n)
    A = (Q * lam) @ Q.conj().T
    out, status = eng.eig([A])
    assert not (status & 3).any()
    mu = out[0][0]
    ref = np.linalg.eigvals(A)
    nrm = np.abs(A).sum(axis=1).max()
    _one_to_one(mu, ref, 2e-13 * nrm)
    cl = np.sort(mu[np.abs(mu - (0.7 + 0.2j)) < 1e-8].real)
    assert len(cl) == 50 and np.all(np.diff(cl) > 5e-11)             # fifty distinct roots, none doubled, none lost


def test_defective_fourfold_eigenvalue(eng):
    """A Jordan block of size 4 hidden by a unitary similarity: an unreduced Hessenberg form exists (the matrix is
    non-derogatory), the eigenvalue is defective, and ANY backward-stable method returns four values on a circle of radius
    ~ (eps ||A||)^(1/4) around it.  Both solvers must put exactly four roots there and agree on the other 116."""
    rng = np.random.default_rng(22)
    n = 120
    lam0 = 0.4 - 0.3j
    D = np.diag(np.concatenate([[lam0] * 4, np.exp(2j * np.pi * rng.random(n - 4)) * (0.5 + 0.5 * rng.random(n - 4))])).astype(complex)
    for i in range(3):
        D[i, i + 1] = 1.0
    Q = _unitary(rng, n)
    A = Q @ D @ Q.conj().T
    out, status = eng.eig([A])
    assert not (status & 3).any()
    mu = out[0][0]
    ref = np.linalg.eigvals(A)
    nrm = np.abs(A).sum(axis=1).max()
    rad = 50 * (np.finfo(float).eps * nrm) ** 0.25
    near, near_ref = np.abs(mu - lam0) < rad, np.abs(ref - lam0) < rad
    assert near.sum() == 4 and near_ref.sum() == 4
    assert abs(mu[near].mean() - lam0) < 1e-10 * nrm                 # the cluster's centre is well conditioned
    _one_to_one(mu[~near], ref[~near_ref], 1e-11 * nrm)


def test_reduced_kbdm_matrices_of_noise_free_members_at_512(eng):
    """The reduced matrix W of noise-free KBDM members at l = m = 512 (formed by the oracle's LAPACK SVD: 16 genuine
    eigenvalues inside the unit disc, 496 spurious ones that crowd the unit circle): the genuine eigenvalues to 1e-11, all
    of them within the spread LAPACK's own backward error allows, in both directions, no doubled root."""
    from oracle import kbdm_oracle as O
    sig = O.brain_sim_signal(2048)
    mats = []
    for m in (512, 448):
        U0, _, U1 = O.compute_U_matrices(sig, m, 1)                                  # kbdm.py:95-130
        L, s, Rh = sla.svd(U0)                                                       # kbdm.py:166
        dsqi = 1.0 / np.sqrt(s)
        W = (dsqi[:, None] * (L.conj().T @ U1 @ Rh.conj().T)) * dsqi[None, :]
        mats.append(W)
    out, status = eng.eig(mats)
    assert not (status & 3).any()
    nfb = eng.last_eig_fallbacks()
    for W, (mu, Pm) in zip(mats, out):
        ref = np.linalg.eigvals(W)
        d = np.abs(mu[:, None] - ref[None, :])
        # eigenvalue sensitivity: LAPACK's own answer moves by eps ||W|| kappa under rounding; take the residual-based bound
        nrm = np.abs(W).sum(axis=1).max()
        res = np.abs(W @ Pm - Pm * mu[None, :]).max(axis=0) / np.abs(Pm).max(axis=0)
        assert res.max() < 1e-11 * nrm * W.shape[0], res.max()                       # every (mu, p) is an eigenpair of W
        inside = np.abs(ref) < 0.999                                                 # the genuine lines decay: |mu| < 1
        assert inside.sum() >= 16
        assert d[:, inside].min(axis=0).max() < 1e-9                                 # each of them found ...
        # ... and no root doubled: a one-to-one matching exists within the spread of the spurious eigenvalues
        _one_to_one(mu, ref, 1e-5)
    assert nfb in (0, 1, 2)
