"""numpy prototype of the eigenvalue solver that kb_aberth.hpp implements: all eigenvalues of an upper Hessenberg
matrix by Ehrlich-Aberth iteration on Hyman's recurrence, started from the eigenvalues of the two diagonal halves
(divide and conquer: a half is again Hessenberg), leaves by a dense solver.  Every root is a column of the
recurrence X (rows from the bottom up), so one iteration for all roots of a node is a triangular-solve-shaped
matrix product H X - the part the device runs on FP64 MFMA.

Counts iterations, checks the result against LAPACK and the a-posteriori tests (power sums) that decide whether a
member falls back to the QR iteration."""
import sys
import numpy as np
import scipy.linalg as sl

EPS = np.finfo(float).eps


def hyman_newton(H, z, stats=None):
    """Newton corrections rho / rho' of det(H - z I) for all z (vector) by Hyman's recurrence, blocked like the
    device code (32 rows per block; per-column power-of-two rescaling at block ends)."""
    n, R = H.shape[0], len(z)
    X = np.zeros((n, R), complex)
    Y = np.zeros((n, R), complex)
    X[n - 1] = 1.0
    b = 32
    k_hi = n - 1
    while k_hi > 0:
        k_lo = max(k_hi - b, 0)                  # rows k in (k_lo, k_hi] determine x_{k-1}
        for k in range(k_hi, k_lo, -1):
            s = H[k, k:] @ X[k:] - z * X[k]
            sp = H[k, k:] @ Y[k:] - z * Y[k] - X[k]
            X[k - 1] = -s / H[k, k - 1]
            Y[k - 1] = -sp / H[k, k - 1]
        mx = np.abs(X[k_lo:k_hi]).max(axis=0)
        with np.errstate(divide="ignore"):
            e = np.where(mx > 0, np.floor(np.log2(np.maximum(mx, 1e-300))), 0)
        big = np.abs(e) > 60
        if big.any():
            f = np.where(big, 2.0 ** (-e), 1.0)
            X *= f
            Y *= f
            if stats is not None:
                stats["rescales"] = stats.get("rescales", 0) + 1
        k_hi = k_lo
    rho = H[0, :] @ X - z * X[0]
    rhop = H[0, :] @ Y - z * Y[0] - X[0]
    return rho / rhop


def aberth_node(H, z0, budget=24, stats=None):
    """Aberth iteration for all eigenvalues of H from the starting values z0.  Returns (z, ok, iterations per root)."""
    n = len(z0)
    z = z0.astype(complex).copy()
    hn = np.abs(H).sum(axis=1).max()
    # separate coincident starting values (deterministically)
    z = z * (1.0 + 1e-9 * np.exp(2j * np.pi * (np.arange(n) * 0.61803398875))) + 1e-12 * hn * np.exp(2j * np.pi * np.arange(n) * 0.754877666)
    active = np.ones(n, bool)
    its = np.zeros(n, int)
    last = np.full(n, np.inf)
    for it in range(budget):
        idx = np.nonzero(active)[0]
        N = hyman_newton(H, z[idx], stats)
        diff = z[idx, None] - z[None, :]
        diff[np.arange(len(idx)), idx] = 1.0
        S = (1.0 / diff).sum(axis=1) - 1.0
        corr = N / (1.0 - N * S)
        bad = ~np.isfinite(corr)
        corr[bad] = 0.0
        z[idx] -= corr
        its[idx] += 1
        ac = np.abs(corr)
        last[idx] = np.where(bad, np.inf, ac)
        conv = ac <= 4 * EPS * np.maximum(np.abs(z[idx]), 1e-6 * hn)
        active[idx[conv & ~bad]] = False
        if not active.any():
            break
    ok = bool(np.all(last <= 1e-10 * np.maximum(np.abs(z), 1e-6 * hn)))
    return z, ok, its


def eig_dc(H, leaf=32, stats=None):
    n = H.shape[0]
    if n <= leaf:
        return np.linalg.eigvals(H), True
    mid = n // 2
    z1, ok1 = eig_dc(H[:mid, :mid], leaf, stats)
    z2, ok2 = eig_dc(H[mid:, mid:], leaf, stats)
    z, ok, its = aberth_node(H, np.concatenate([z1, z2]), stats=stats)
    if stats is not None:
        stats.setdefault("levels", []).append((n, float(its.mean()), int(its.max())))
        stats["work"] = stats.get("work", 0.0) + float(its.sum()) * n * n      # root-iterations x n^2 (x 8 real flops)
    # a-posteriori: the first two power sums against the traces
    t1, t2 = np.trace(H), np.trace(H @ H)
    sc = np.abs(z).sum() + 1e-300
    ok = ok and ok1 and ok2 and abs(z.sum() - t1) <= 1e-9 * sc and abs((z * z).sum() - t2) <= 1e-9 * (np.abs(z) ** 2).sum()
    return z, ok


def check(W, name):
    H = sl.hessenberg(np.array(W, complex))
    n = H.shape[0]
    st = {}
    z, ok = eig_dc(H, stats=st)
    ref = np.linalg.eigvals(W)
    d = np.abs(z[:, None] - ref[None, :])
    err = d.min(axis=1).max() / max(np.abs(ref).max(), 1e-300)
    once = len(set(d.argmin(axis=1))) == n
    lv = " ".join(f"{a}:{b:.1f}/{c}" for a, b, c in st.get("levels", [])[-3:])
    print(f"{name:34s} n={n:4d} ok={ok} err {err:.1e} once {once} work {st.get('work', 0) / n ** 3:5.1f} n^3 (x8 flops) rescales {st.get('rescales', 0)} | top levels {lv}", flush=True)
    return ok and once and err < 1e-9


if __name__ == "__main__":
    sys.path.insert(0, ".")
    sys.path.insert(0, "tools")
    from oracle import kbdm_oracle as O
    rng = np.random.default_rng(0)

    def reduced(sig, m, l=None, p=1, q=0.0):
        U0, Up1, Up = O.compute_U_matrices(sig, m, p)
        L, s, Rh = sl.svd(Up1)
        l = l or m
        d = 1 / np.sqrt(s[:l] + (q * q / s[:l] if q else 0))
        return (d[:, None] * (L[:, :l].conj().T @ Up @ Rh[:l].conj().T)) * d[None, :]

    allok = True
    for n in (40, 100, 257, 400):
        allok &= check(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)), "random complex")
    allok &= check(rng.standard_normal((200, 200)), "random real")
    for sigma, m, N in ((1e-3, 150, 2048), (1e-3, 400, 2048), (1e-6, 256, 2048), (0.0, 300, 1024), (0.0, 150, 1024), (1e-2, 200, 2048)):
        sig = O.brain_sim_signal(N)
        if sigma:
            sig = O.make_noisy(sig, sigma, 1)
        allok &= check(reduced(sig, m), f"kbdm sigma={sigma} m={m}")
    sig = O.make_noisy(O.brain_sim_signal(2048), 1e-3, 2)
    allok &= check(reduced(sig, 180, l=30), "kbdm m=180 l=30")
    allok &= check(reduced(sig, 256, q=1e-3), "kbdm m=256 q=1e-3")
    allok &= check(reduced(sig, 200, p=2), "kbdm m=200 p=2")
    # structured trouble: nearly decoupled, defective, multiple eigenvalues
    A = rng.standard_normal((120, 120)) + 0j
    A[60:, :60] = 0
    allok &= check(A, "block triangular (exact split)")
    J = np.diag(np.ones(80), 0) + np.diag(np.ones(79), 1)
    allok &= check(J + 1e-3 * rng.standard_normal((80, 80)), "perturbed Jordan block")
    allok &= check(np.diag(np.repeat(np.arange(1.0, 31), 3)) + 1e-13 * rng.standard_normal((90, 90)), "triple eigenvalues")
    print("ALL OK" if allok else "SOME FAILED")
