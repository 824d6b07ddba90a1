"""numpy prototype of the bidiagonal divide-and-conquer SVD that kb_bdsdc.hpp implements (own derivation in the
organisation of LAPACK dbdsdc / dlasd0-4; written to fix the algorithm and its numerics before the HIP version).

B upper bidiagonal n x n (d, e)  ->  B = X diag(s) Y^T,  s descending.
Node = rows [lo, hi), columns [lo, hi + sqre).  Leaves: one-sided Jacobi.  Merge: secular equation + Loewner z +
ONE dense coefficient matrix per side (deflation rotations, the permutation and the null-column rotation folded into it),
so that the new vectors are  Ubasis @ CU  and  Vbasis @ CV  with block-diagonal bases.
"""
import numpy as np

EPS = np.finfo(float).eps / 2      # LAPACK dlamch('E')


def jacobi_leaf(A):
    """A: r x c (c = r or r + 1).  Returns U (r x r), s (r, ascending), V (c x c); for c = r + 1 the last column of V
    is the null vector.  One-sided Jacobi on the columns of A."""
    r, c = A.shape
    W = A.astype(float).copy()
    V = np.eye(c)
    for sweep in range(60):
        rotated = False
        for p in range(c - 1):
            for q in range(p + 1, c):
                a = W[:, p] @ W[:, p]
                b = W[:, q] @ W[:, q]
                g = W[:, p] @ W[:, q]
                if abs(g) <= EPS * np.sqrt(a * b) or g == 0.0:
                    continue
                rotated = True
                zeta = (b - a) / (2.0 * g)
                t = np.sign(zeta) / (abs(zeta) + np.hypot(1.0, zeta)) if zeta != 0 else 1.0
                cs = 1.0 / np.hypot(1.0, t)
                sn = cs * t
                for M in (W, V):
                    x, y = M[:, p].copy(), M[:, q].copy()
                    M[:, p] = cs * x - sn * y
                    M[:, q] = sn * x + cs * y
        if not rotated:
            break
    nrm = np.sqrt((W * W).sum(axis=0))
    order = np.argsort(nrm, kind="stable")          # ascending; for c = r + 1 the smallest is the null column
    if c > r:
        null, order = order[0], order[1:]
    s = nrm[order]
    U = np.zeros((r, r))
    for k, j in enumerate(order):
        U[:, k] = W[:, j] / s[k] if s[k] > 0 else 0.0
    # complete U for (numerically) zero singular values: Gram-Schmidt of unit vectors
    for k in range(r):
        if not (s[k] > np.finfo(float).tiny):
            for e in range(r):
                v = np.zeros(r); v[e] = 1.0
                for _ in range(2):
                    for j in range(r):
                        if j != k and (s[j] > np.finfo(float).tiny or j < k):
                            v -= (U[:, j] @ v) * U[:, j]
                nv = np.linalg.norm(v)
                if nv > 0.5:
                    U[:, k] = v / nv
                    break
    Vout = V[:, list(order) + ([null] if c > r else [])]
    return U, s, Vout


def secular_root(i, K, d, z2, rho_inv, stats=None):
    """Root i (0-based) of  g(x) = rho_inv + sum_j z2_j / (d_j^2 - x),  x = sigma^2, d ascending with d[0] = 0.
    Returns (origin index o, tau) with sigma = d[o] + tau, plus delta_j = d_j - sigma, w_j = d_j + sigma."""
    last = i == K - 1
    if not last:
        dl, dr = d[i], d[i + 1]
        gap = dr - dl
        mid = 0.5 * gap
        # g at the midpoint decides the origin
        sig = dl + mid
        delta = (d - dl) - mid
        w = (d + dl) + mid
        gm = rho_inv + np.sum(z2 / (delta * w))
        if gm > 0:
            o, lo, hi = i, 0.0, mid                    # root in (dl, mid]
        else:
            o, lo, hi = i + 1, -mid, 0.0               # root in [mid, dr)
        tau = mid if o == i else -mid
    else:
        o = K - 1
        # g(x) > 0 for x >= d^2 + rho*|z|^2: sigma < sqrt(d^2 + rho)  (|z| = 1)
        rho = 1.0 / rho_inv
        up = rho / (d[o] + np.sqrt(d[o] * d[o] + rho))
        lo, hi = 0.0, up
        tau = 0.5 * up
    do = d[o]
    for it in range(80):
        delta = (d - do) - tau
        w = (d + do) + tau
        t = z2 / (delta * w)                           # terms z_j^2 / (d_j^2 - x)
        dt = t / (delta * w)                           # derivative terms wrt x
        if last:
            psi, dpsi, phi, dphi = t.sum(), dt.sum(), 0.0, 0.0
        else:
            psi, dpsi = t[:i + 1].sum(), dt[:i + 1].sum()
            phi, dphi = t[i + 1:].sum(), dt[i + 1:].sum()
        g = rho_inv + psi + phi
        err = 8.0 * (abs(psi) + abs(phi)) + rho_inv + abs(tau) * (dpsi + dphi) * 0   # |terms| bound
        if g == 0.0 or abs(g) <= 4 * EPS * (np.abs(t).sum() + rho_inv):
            break
        if g < 0:
            lo = tau
        else:
            hi = tau
        sig = do + tau
        # rational model in x with poles p_i, p_{i+1} (distances a = p_i - x < 0, b = p_{i+1} - x > 0)
        a = delta[i] * w[i]
        if not last:
            b = delta[i + 1] * w[i + 1]
            S, s0 = dpsi * a * a, psi - dpsi * a
            R, r0 = dphi * b * b, phi - dphi * b
            c = rho_inv + s0 + r0
            # c + S/(a - eta) + R/(b - eta) = 0  ->  c eta^2 - (c(a+b) + S + R) eta + (c a b + S b + R a) = 0
            qa, qb, qc = c, -(c * (a + b) + S + R), c * a * b + S * b + R * a
            disc = qb * qb - 4 * qa * qc
            eta = None
            if disc >= 0:
                sq = np.sqrt(disc)
                # both roots; take the one with a < eta < b
                cands = []
                if qa != 0:
                    q_ = -0.5 * (qb + np.copysign(sq, qb))
                    cands = [q_ / qa, (qc / q_) if q_ != 0 else np.inf]
                elif qb != 0:
                    cands = [-qc / qb]
                cands = [x for x in cands if a < x < b]
                if cands:
                    eta = min(cands, key=abs)
        else:
            # two-pole model with p_{K-2}, p_{K-1} is LAPACK's choice; the one-pole model + safeguard is enough here
            S, s0 = dpsi * a * a, psi - dpsi * a
            c = rho_inv + s0
            eta = a + S / c if c > 0 else None         # c + S/(a - eta) = 0
        new_tau = None
        if eta is not None and np.isfinite(eta):
            x2 = sig * sig + eta
            if x2 > 0:
                dsg = eta / (sig + np.sqrt(x2))
                cand = tau + dsg
                if lo < cand < hi:
                    new_tau = cand
        if new_tau is None:
            new_tau = 0.5 * (lo + hi)
        if new_tau == tau or not (lo < new_tau < hi):
            break
        tau = new_tau
    if stats is not None:
        stats.append(it)
    return o, tau


def merge(alpha, beta, U1, D1, V1, U2, D2, V2, sqre, descending=False, stats=None):
    """One merge.  Children: (U1 nl x nl, D1, V1 (nl+1)^2 with null column last), (U2, D2, V2 (nr+sqre)^2).
    Returns U (n x n), D (n), V (n+sqre)^2 with the null column last (sqre = 1)."""
    nl, nr = len(D1), len(D2)
    n = nl + 1 + nr
    mcols = n + sqre
    # ---- the z row in the basis [V1a | v1 | V2a | v2]
    l1, lam1 = V1[nl, :nl], V1[nl, nl]
    f2 = V2[0, :nr]
    phi2 = V2[0, nr] if sqre else 0.0
    z = np.zeros(n)
    dd = np.zeros(n)
    # local index j: 0 = the special (center row / q column), 1..nl = child 1, nl+1.. = child 2
    z[1:nl + 1] = alpha * l1
    z[nl + 1:] = beta * f2
    dd[1:nl + 1] = D1
    dd[nl + 1:] = D2
    a1, b2 = alpha * lam1, beta * phi2
    if sqre:
        r0 = np.hypot(a1, b2)
        c0, s0 = (a1 / r0, b2 / r0) if r0 > 0 else (1.0, 0.0)
        z[0] = r0
    else:
        c0, s0 = 1.0, 0.0
        z[0] = a1
    # U basis column j: j = 0 -> e_center (row nl); 1..nl -> U1 col j-1 (rows 0..nl-1); nl+1.. -> U2 col (rows nl+1..)
    # V basis column j: 0 -> q = c0 v1 + s0 v2 (or v1); 1..nl -> V1a; nl+1.. -> V2a; n (sqre) -> null = -s0 v1 + c0 v2
    # Coefficient matrices over the RAW bases  Ub = [U1 . .; . 1 .; . . U2] (cols: U1 (nl), center, U2 (nr)) and
    # Vb = [V1 .; . V2] (cols: V1a (nl), v1, V2a (nr), v2 (sqre)):  TU (n x n), TV (mcols x mcols) map local index -> raw
    ub_col = np.zeros(n, dtype=int)
    ub_col[0] = nl
    ub_col[1:nl + 1] = np.arange(nl)
    ub_col[nl + 1:] = nl + 1 + np.arange(nr)
    GU = np.zeros((n, n))                 # raw-basis coefficients of the (rotated) local U basis vectors
    GU[ub_col, np.arange(n)] = 1.0
    GV = np.zeros((mcols, mcols))
    vb_col = np.zeros(n, dtype=int)
    vb_col[1:nl + 1] = np.arange(nl)
    vb_col[nl + 1:] = nl + 1 + np.arange(nr)
    GV[vb_col[1:], np.arange(1, n)] = 1.0
    GV[nl, 0] = c0
    if sqre:
        GV[nl + 1 + nr, 0] = s0
        GV[nl, n] = -s0
        GV[nl + 1 + nr, n] = c0
    # ---- scale
    org = max(abs(alpha), abs(beta), dd.max() if n > 1 else 0.0)
    if org == 0.0:
        org = 1.0
    dd = dd / org
    z = z / org
    tol = 8.0 * EPS * max(abs(alpha) / org, abs(beta) / org, dd.max())
    # ---- sort (index 0 stays first), deflate
    order = [0] + list(1 + np.argsort(dd[1:], kind="stable"))
    if abs(z[0]) <= tol:
        z[0] = tol
    keep, defl = [0], []
    prev = None
    for j in order[1:]:
        if abs(z[j]) <= tol:
            defl.append(j)
            continue
        if prev is not None and dd[j] - dd[prev] <= tol:
            # rotate (prev, j): z_prev -> 0
            s_, c_ = z[prev], z[j]
            tau_ = np.hypot(c_, s_)
            c_, s_ = c_ / tau_, -s_ / tau_
            z[j], z[prev] = tau_, 0.0
            for G in (GU, GV):
                x, y = G[:, prev].copy(), G[:, j].copy()
                G[:, prev] = c_ * x + s_ * y           # drot(x = prev, y = j, c, s): x' = c x + s y, y' = c y - s x
                G[:, j] = c_ * y - s_ * x
            defl.append(prev)
            keep.remove(prev)
        keep.append(j)
        prev = j
    K = len(keep)
    dk = dd[keep].copy()
    zk = z[keep].copy()
    if K > 1 and dk[1] <= tol / 2:
        dk[1] = tol / 2
    # ---- secular equation
    rho = zk @ zk
    zn = zk / np.sqrt(rho)
    z2 = zn * zn
    sig = np.zeros(K)
    DEL = np.zeros((K, K))            # DEL[j, i] = dk_j - sigma_i
    SUM = np.zeros((K, K))
    for i in range(K):
        o, tau = secular_root(i, K, dk, z2, 1.0 / rho, stats)
        sig[i] = dk[o] + tau
        DEL[:, i] = (dk - dk[o]) - tau
        SUM[:, i] = (dk + dk[o]) + tau
    # ---- Loewner z
    zh = np.zeros(K)
    for j in range(K):
        v = DEL[j, K - 1] * SUM[j, K - 1]
        for i in range(j):
            v *= DEL[j, i] * SUM[j, i] / (dk[j] - dk[i]) / (dk[j] + dk[i])
        for i in range(j, K - 1):
            v *= DEL[j, i] * SUM[j, i] / (dk[j] - dk[i + 1]) / (dk[j] + dk[i + 1])
        zh[j] = np.copysign(np.sqrt(abs(v)), zk[j])
    # ---- vectors of the K x K problem
    VM = zh[:, None] / (DEL * SUM)
    UM = dk[:, None] * VM
    UM[0, :] = -1.0
    VM /= np.linalg.norm(VM, axis=0)
    UM /= np.linalg.norm(UM, axis=0)
    # ---- all n singular values, output order
    vals = np.concatenate([sig, dd[defl]]) * org
    src = [("k", i) for i in range(K)] + [("d", j) for j in defl]
    perm = np.argsort(-vals if descending else vals, kind="stable")
    CU = np.zeros((n, n))
    CV = np.zeros((mcols, mcols))
    D = np.zeros(n)
    for newc, p in enumerate(perm):
        D[newc] = vals[p]
        kind, idx = src[p]
        if kind == "k":
            CU[:, newc] = GU[:, keep] @ UM[:, idx]
            CV[:, newc] = GV[:, keep] @ VM[:, idx]
        else:
            CU[:, newc] = GU[:, idx]
            CV[:, newc] = GV[:, idx]
    if sqre:
        CV[:, n] = GV[:, n]
    Ub = np.zeros((n, n))
    Ub[:nl, :nl] = U1
    Ub[nl, nl] = 1.0
    Ub[nl + 1:, nl + 1:] = U2
    Vb = np.zeros((mcols, mcols))
    Vb[:nl + 1, :nl + 1] = V1
    Vb[nl + 1:, nl + 1:] = V2
    return Ub @ CU, D, Vb @ CV


def bdsdc(d, e, leaf=16, stats=None):
    d, e = np.asarray(d, float), np.asarray(e, float)
    n = len(d)

    def solve(lo, hi, sqre, top):
        nn = hi - lo
        if nn <= leaf:
            A = np.zeros((nn, nn + sqre))
            for r in range(nn):
                A[r, r] = d[lo + r]
                if r + 1 < nn + sqre:
                    A[r, r + 1] = e[lo + r]
            U, s, V = jacobi_leaf(A)
            if top:
                o = np.argsort(-s, kind="stable")
                return U[:, o], s[o], V[:, o]
            return U, s, V
        nl = (nn - 1) // 2
        ic = lo + nl
        U1, D1, V1 = solve(lo, ic, 1, False)
        U2, D2, V2 = solve(ic + 1, hi, sqre, False)
        return merge(d[ic], e[ic] if ic < len(e) else 0.0, U1, D1, V1, U2, D2, V2, sqre, descending=top, stats=stats)
    return solve(0, n, 0, True)


def bidiag(A):
    """Householder bidiagonalisation of a complex square matrix to REAL upper bidiagonal (d, e) (zgebrd's form)."""
    A = np.array(A, dtype=complex)
    n = A.shape[0]
    d, e = np.zeros(n), np.zeros(n - 1)

    def house(x):
        alpha = x[0]
        xn = np.linalg.norm(x[1:])
        if xn == 0 and alpha.imag == 0:
            return np.zeros_like(x), 0.0, alpha.real
        beta = -np.copysign(np.hypot(abs(alpha), xn), alpha.real)
        tau = (beta - alpha) / beta
        v = x / (alpha - beta)
        v[0] = 1.0
        return v, tau, beta
    for k in range(n):
        v, tau, beta = house(A[k:, k].copy())
        A[k:, k:] -= np.conj(tau) * np.outer(v, v.conj() @ A[k:, k:])      # H^H A
        d[k] = beta
        if k < n - 1:
            v, tau, beta = house(A[k, k + 1:].conj().copy())
            A[k:, k + 1:] -= tau * np.outer(A[k:, k + 1:] @ v, v.conj())
            e[k] = beta
    return d, e


if __name__ == "__main__":
    import sys
    sys.path.insert(0, ".")
    from oracle import kbdm_oracle as O
    import scipy.linalg as sl
    rng = np.random.default_rng(0)

    def check(dv, ev, name):
        n = len(dv)
        B = np.diag(dv) + np.diag(ev, 1)
        st = []
        X, s, Y = bdsdc(dv, ev, stats=st)
        sref = np.linalg.svd(B, compute_uv=False)
        print(f"{name:28s} n={n:4d} sv {np.abs(s - sref).max() / sref[0]:.1e}  res {np.abs(X * s @ Y.T - B).max() / sref[0]:.1e}"
              f"  orthX {np.abs(X.T @ X - np.eye(n)).max():.1e} orthY {np.abs(Y.T @ Y - np.eye(n)).max():.1e}"
              f"  secular its mean {np.mean(st) if st else 0:.1f} max {max(st) if st else 0}")

    for n in (5, 17, 40, 100, 257):
        check(rng.standard_normal(n), rng.standard_normal(n - 1), "random")
    check(np.ones(64), np.ones(63), "ones")
    check(2.0 ** -np.arange(60), 2.0 ** -np.arange(59), "graded")
    check(np.r_[np.ones(30), 1e-9 * np.ones(30)], 1e-12 * np.ones(59), "clusters")
    check(np.zeros(20), np.ones(19), "zero diagonal")
    check(np.ones(20), np.zeros(19), "identity")
    # the bidiagonal forms of real KBDM Hankel matrices: noise-free (sv down to 1e-16 s0) and noisy
    for sigma, m, N in ((0.0, 150, 1024), (1e-3, 200, 2048), (1e-6, 256, 2048), (0.0, 300, 1024)):
        sig = O.brain_sim_signal(N)
        if sigma:
            sig = O.make_noisy(sig, sigma, 1)
        U0 = sl.hankel(sig[:m], sig[m - 1:2 * m - 1])
        dv, ev = bidiag(U0)
        check(dv, ev, f"hankel sigma={sigma} m={m}")
