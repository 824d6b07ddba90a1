"""numpy model of the eigenvalue-only multishift QR iteration of kb_hqr2.hpp (active block, zlahqr deflation test, 2 nb
shifts per sweep from the trailing block, exceptional shifts every 6th sweep without deflation) with and without an
aggressive early deflation step built on LEFT EIGENVECTORS of the trailing window instead of its Schur form:

  window T = H[kw:i+1, kw:i+1], spike beta = H[kw, kw-1];  eigenvalues of T (on the device: the Ehrlich-Aberth solver
  that computes the shifts anyway);  left eigenvectors by Hyman's recurrence (y_0 = 1);  a vector is deflatable when
  |beta| |q_0| <= tol for its orthonormalised form q;  Z = [Z1 | Q2] unitary with span(Q2) = the accepted vectors:
  Z^H T Z = [A11 A12; E A22] with E = O(eps) (left-invariant subspace) - verified, then dropped with A12, A22 (eigenvalues
  only: A22's eigenvalues are the accepted lambdas);  A11 with the new spike beta Z1[0,:]^H is reduced back to
  Hessenberg form;  the strip above the window is multiplied by Z1.

Counts sweeps / shifts per eigenvalue on reduced KBDM matrices.  Written to decide whether the step pays before
building it for the device.
"""
import sys
import numpy as np
import scipy.linalg as sl

ULP = np.finfo(float).eps
SMLNUM = np.finfo(float).tiny


def cabs1(z):
    return abs(z.real) + abs(z.imag)


def small_subdiag(H, k, n, smlnum):
    h = H[k, k - 1]
    if cabs1(h) <= smlnum:
        return True
    tst = cabs1(H[k - 1, k - 1]) + cabs1(H[k, k])
    if tst == 0:
        if k - 2 >= 0:
            tst += cabs1(H[k - 1, k - 2])
        if k + 1 <= n - 1:
            tst += cabs1(H[k + 1, k])
    if cabs1(h) <= ULP * tst:
        a1, a2 = cabs1(h), cabs1(H[k - 1, k])
        ab, ba = max(a1, a2), min(a1, a2)
        df = H[k - 1, k - 1] - H[k, k]
        b1, b2 = cabs1(H[k, k]), cabs1(df)
        aa, bb = max(b1, b2), min(b1, b2)
        s = aa + ab
        if ba * (ab / s) <= max(smlnum, ULP * (bb * (aa / s))):
            return True
    return False


def left_vectors(T, lams):
    """Hyman's recurrence for the left null vectors of T - lam I, y_0 = 1; returns Y (columns) and the residual of the
    last equation relative to |y|."""
    nw = T.shape[0]
    Y = np.zeros((nw, len(lams)), complex)
    res = np.zeros(len(lams))
    for k, lam in enumerate(lams):
        y = np.zeros(nw, complex)
        y[0] = 1.0
        A = T - lam * np.eye(nw)
        for j in range(nw - 1):
            s = np.conj(y[:j + 1]) @ A[:j + 1, j]
            y[j + 1] = np.conj(-s / A[j + 1, j])
            if abs(y[j + 1]) > 1e100:
                y /= 1e100
        r = np.conj(y) @ A[:, nw - 1]
        Y[:, k] = y
        res[k] = abs(r) / np.linalg.norm(y)
    return Y, res


def aed_step(H, l, i, nw, stats, check=True):
    """One deflation attempt on the active block [l, i].  Returns (number deflated, eigenvalues deflated, shifts)."""
    n = H.shape[0]
    kw = i - nw + 1
    beta = H[kw, kw - 1] if kw > l else 0.0
    T = H[kw:i + 1, kw:i + 1].copy()
    lams = np.linalg.eigvals(T)
    if kw == l:
        return nw, lams, None
    Y, res = left_vectors(T, lams)
    nrm = np.linalg.norm(Y, axis=0)
    crit = abs(beta) * np.abs(Y[0, :]) / nrm
    order = np.argsort(crit)
    Q = np.zeros((nw, 0), complex)
    acc = []
    tnorm = np.abs(T).sum(axis=1).max()
    for k in order:
        tol = max(SMLNUM, ULP * abs(lams[k]))
        if crit[k] > tol * 4:
            break
        v = Y[:, k] / nrm[k]
        for _ in range(2):
            v = v - Q @ (Q.conj().T @ v)
        nv = np.linalg.norm(v)
        if nv < 1e-3:
            continue
        v = v / nv
        if abs(beta) * abs(v[0]) > tol:
            continue
        # the row of E this vector would leave:  v^H T (I - [Q v][Q v]^H)
        Qn = np.column_stack([Q, v])
        e = v.conj() @ T - (v.conj() @ T @ Qn) @ Qn.conj().T
        if np.abs(e).max() > ULP * max(abs(lams[k]), 1e-300) * 4 and np.abs(e).max() > ULP * tnorm:
            stats["e_reject"] = stats.get("e_reject", 0) + 1
            continue
        Q = Qn
        acc.append(k)
    kd = len(acc)
    shifts = np.delete(lams, acc)
    if kd == 0:
        return 0, [], shifts
    # unitary Z = [Z1 | Q2]: Householder QR of Q
    Zfull, _ = np.linalg.qr(Q, mode="complete")
    Z1 = Zfull[:, kd:]
    ns = nw - kd
    A11 = Z1.conj().T @ T @ Z1
    spike = beta * Z1[0, :].conj()
    # [spike | A11] back to Hessenberg: reflector taking spike to a multiple of e_1, then Hessenberg reduction
    if ns > 0:
        x = spike.copy()
        alpha = x[0]
        nx = np.linalg.norm(x)
        if nx > 0:
            ph = alpha / abs(alpha) if abs(alpha) > 0 else 1.0
            v = x.copy()
            v[0] += ph * nx
            v /= np.linalg.norm(v)
            P = np.eye(ns) - 2.0 * np.outer(v, v.conj())
            A11 = P.conj().T @ A11 @ P
            Z1 = Z1 @ P
            spike = P.conj().T @ spike
        Hh, Qh = sl.hessenberg(A11, calc_q=True)
        Z1 = Z1 @ Qh
        H[kw:kw + ns, kw:kw + ns] = Hh
        H[kw, kw - 1] = spike[0]
        H[kw + 1:i + 1, kw - 1] = 0
        H[l:kw, kw:kw + ns] = H[l:kw, kw:i + 1] @ Z1
    return kd, lams[acc], shifts


def hqr_model(W, nb=8, aed=0, verbose=False):
    """Eigenvalues of W by the modelled iteration; aed = window size (0: none).  Returns (eigs, stats)."""
    H = sl.hessenberg(np.array(W, complex))
    n = H.shape[0]
    smlnum = SMLNUM * (n / ULP)
    w = np.zeros(n, complex)
    stats = {"sweeps": 0, "shifts": 0, "aed_calls": 0, "aed_defl": 0, "aed_skip_sweep": 0, "work": 0.0}
    i = n - 1
    while i >= 0:
        l = 0
        kdefl = 0
        while True:
            kf = l
            for k in range(i, l, -1):
                if small_subdiag(H, k, n, smlnum):
                    kf = k
                    break
            l = kf
            if l > 0:
                H[l, l - 1] = 0
            if l >= i:
                w[i] = H[i, i]
                i = l - 1
                break
            na = i - l + 1
            if na < 9:
                w[l:i + 1] = np.linalg.eigvals(H[l:i + 1, l:i + 1])
                i = l - 1
                break
            kdefl += 1
            nbb = max(1, min(nb, na // 6))
            ns = 2 * nbb
            sh = None
            if aed and na >= aed + 8:
                stats["aed_calls"] += 1
                kd, ev, shifts = aed_step(H, l, i, aed, stats)
                if kd > 0:
                    stats["aed_defl"] += kd
                    w[i - kd + 1:i + 1] = ev
                    i -= kd
                    kdefl = 0
                    if kd >= max(2, aed // 8):          # enough progress: look again before sweeping (LAPACK's "nibble")
                        stats["aed_skip_sweep"] += 1
                        continue
                    na = i - l + 1
                    if na < 9:
                        continue
                    nbb = max(1, min(nb, na // 6))
                    ns = 2 * nbb
                if shifts is not None and len(shifts) >= ns:
                    # the undeflated window eigenvalues are the shifts (those closest to the bottom entry first)
                    sh = shifts[np.argsort(np.abs(shifts - H[i, i]))][:ns]
            if sh is None:
                if kdefl % 6 == 0:
                    sh = np.array([H[i - (b & ~1), i - (b & ~1)] + 0.75 * cabs1(H[i - (b & ~1), i - (b & ~1) - 1]) for b in range(ns)])
                else:
                    sh = np.linalg.eigvals(H[i - ns + 1:i + 1, i - ns + 1:i + 1])
            A = H[l:i + 1, l:i + 1]
            for s in sh:
                Qm, Rm = np.linalg.qr(A - s * np.eye(na))
                A = Rm @ Qm + s * np.eye(na)
                A = np.triu(A, -1)
            H[l:i + 1, l:i + 1] = A
            stats["sweeps"] += 1
            stats["shifts"] += len(sh)
            stats["work"] += len(sh) * na * na
    stats["shifts_per_eig"] = stats["shifts"] / n
    return w, stats


def reduced_kbdm_matrix(m, N=2048, sigma=1e-3, seed=0):
    sys.path.insert(0, ".")
    from oracle import kbdm_oracle as O
    sig = O.brain_sim_signal(N)
    if sigma:
        sig = O.make_noisy(sig, sigma, seed)
    U0, Up1, Up = O.compute_U_matrices(sig, m, 1)
    L, s, Rh = sl.svd(Up1)
    D = np.diag(1 / np.sqrt(s))
    return D @ L.conj().T @ Up @ Rh.conj().T @ D


if __name__ == "__main__":
    for m in (120, 200):
        W = reduced_kbdm_matrix(m)
        ref = np.linalg.eigvals(W)
        for aed in (0, 24, 32, 48):
            w, st = hqr_model(W, aed=aed)
            err = np.abs(w[:, None] - ref[None, :]).min(axis=1).max()
            once = len(set(np.abs(w[:, None] - ref[None, :]).argmin(axis=1))) == m
            print(f"m={m} aed={aed:2d}: sweeps {st['sweeps']:4d} shifts/eig {st['shifts_per_eig']:.2f} work {st['work']/m**3:.2f} n^3 "
                  f"aed calls {st['aed_calls']} deflated {st['aed_defl']} skipped sweeps {st['aed_skip_sweep']} e_reject {st.get('e_reject', 0)} | err {err:.1e} all-once {once}",
                  flush=True)
