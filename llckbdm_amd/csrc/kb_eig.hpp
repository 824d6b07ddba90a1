// Per-ensemble-member dense nonsymmetric complex eigen-solver (replaces scipy.linalg.eig
// at reference kbdm.py:192, LAPACK zgeev there).  One workgroup owns one n x n matrix.
//
// Only (mu_k, p_k) pairs are consumed downstream (kbdm.py:198-236: each eigenvector is
// normalised on its own and its overall scale cancels), so instead of a full Schur form
// with accumulated Schur vectors we compute
//   1. Householder Hessenberg reduction  W = Qh H Qh^H               (gehd2 / gen_qh)
//   2. eigenvalues of H by single-shift complex QR on the ACTIVE block only
//      (LAPACK zlahqr with wantt = wantz = false)                      (hqr_eigvals)
//   3. one eigenvector per eigenvalue by inverse iteration on H (LAPACK zlaein: LU of
//      H - w I with adjacent-row pivoting + one triangular solve), one wavefront per
//      eigenvalue, embarrassingly parallel                             (invit)
//   4. back-transformation p_k = Qh x_k (done by the batched zgemm kernel).
// Column-major storage, leading dimension ld.
#pragma once
#include "kb_ctx.hpp"
#include "kb_svd.hpp"

namespace kb {

KB_HD int gehd2_scratch_bytes(int n, int nwaves, int ws) {
    int z = n > nwaves * ws ? n : nwaves * ws;
    return (n + z) * (int)sizeof(cd);
}

// Hessenberg reduction: W = Qh H Qh^H, Qh = H_0 ... H_{n-3}; reflector k stored in
// W[k+2.., k] (v[0] = 1 at row k+1); the subdiagonal W[k+1,k] becomes real.
template <class C>
KB_HD void gehd2(const C& ctx, int n, cd* W, int ld, cd* tauh, int k0 = 0) {
    cd* vv = reinterpret_cast<cd*>(ctx.scratch());
    cd* zp = vv + n;
    for (int k = k0; k < n - 2; ++k) {
        cd* col = &W[(k + 1) + (size_t)k * ld];
        const int nv = n - k - 1;
        double beta;
        cd tau;
        larfg(ctx, nv, col, beta, tau);
        ctx.sync();
        // right: W[0:n, k+1:n] -= tau (W v) v^H  with v = [1; col[1..]]
        if (!is_zero(tau)) {
            for (int i = ctx.tid(); i < nv; i += ctx.nthreads()) vv[i] = (i == 0) ? mk(1.0, 0.0) : col[i];
            ctx.sync();
            apply_right(ctx, n, nv, vv, tau, &W[(size_t)(k + 1) * ld], ld, zp);
            ctx.sync();
            // left: W[k+1:n, k+1:n] = (I - conj(tau) v v^H) W[k+1:n, k+1:n]
            apply_left(ctx, nv, nv, col, conj(tau), &W[(k + 1) + (size_t)(k + 1) * ld], ld);
        }
        ctx.sync();
        if (ctx.tid() == 0) { tauh[k] = tau; col[0] = mk(beta, 0.0); }
        ctx.sync();
    }
}

// ---------------------------------------------------------------------------------
// Blocked Hessenberg reduction (zgehrd / zlahr2 organisation).  With Q = H_0 ... H_{j-1} = I - V T V^H the matrix
// after j reflectors of a panel is  A^(j) = Q^H A0 Q,  A0 Q = A0 - Y V^H  with  Y = A0 V T:
//   y_j = tau_j (A0 v_j - Y_j (V_j^H v_j)) ,   T(0:j, j) = -tau_j T_j (V_j^H v_j) ,  T(j, j) = tau_j ,
// so inside a panel of KB_NB columns only the current column is formed explicitly,
//   x = A0(:, k) - Y_j V_j(k, :)^H ,   column k of A^(j) = x - V_j T_j^H (V_j^H x) ,
// and the untouched A0 is streamed ONCE per column (the product A0 v_j, rows below the panel's first row only: the
// rows above follow after the panel, hess_ytop_block; the left factors never need a pass of their own inside the panel).  All columns to the right of the panel then receive ONE rank-2NB update
//   A <- A0 - Y V^H - V Z^H ,   Z = (A0^H V - V (Y^H V)) T = A0^H (V T) - V (Y^H V T)
// (k_hess_z: one more pass over the trailing columns for all NB vectors at once, then k_hess_update on FP64
// MFMA).  The panel leaves VT = V T (N x NB) and MT = (Y^H V) T (NB x NB) for k_hess_z.
// v_t (panel column t, global column kt = p0 + t): rows > kt+1 stored in W[.., kt], implicit 1 at
// row kt+1, zero above.
KB_HD cd hess_vt(const cd* W, int ld, int p0, int r, int t) {
    const int kt = p0 + t;
    return (r > kt + 1) ? W[r + (size_t)kt * ld] : ((r == kt + 1) ? mk(1.0, 0.0) : czero());
}
// (the panel itself: kb_panel_team.hpp, hess_panel_team - teams of T >= 1 workgroups per member)

// Deferred left factor of one panel for the trailing columns c in [c_begin, c_end):
//   Z(c, t) = sum_r conj(A0(r, c)) VT(r, t) - sum_u V(c, u) MT(u, t)          (reference form; the device kernel
// k_hess_z computes the same sums tiled through LDS).
template <class C>
KB_HD void hess_z_block(const C& ctx, int N, const cd* W, int ld, int p0, const cd* VT, int ldvt, const cd* MT,
                        cd* Z, int ldz, int c_begin, int c_end) {
    for (int idx = ctx.tid(); idx < (c_end - c_begin) * KB_NB; idx += ctx.nthreads()) {
        const int c = c_begin + idx % (c_end - c_begin), t = idx / (c_end - c_begin);
        cd acc = czero();
        for (int r = p0 + 1; r < N; ++r) cfmac(acc, W[r + (size_t)c * ld], VT[r + (size_t)t * ldvt]);
        for (int u = 0; u < KB_NB; ++u) acc = acc - hess_vt(W, ld, p0, c, u) * MT[u + t * KB_NB];
        Z[c + (size_t)t * ldz] = acc;
    }
    ctx.sync();
}

// The rows above a panel (r <= p0), which hess_panel_team leaves alone: their share of Y = A0 V T and their entries in the
// panel's own columns,
//   Y(r, t) = sum_{c > p0} A0(r, c) VT(c, t) ,      W(r, p0 + j) -= sum_{t < j} Y(r, t) conj(V(p0 + j, t)) ,  j = 1 .. NB - 1
// (zlahr2 finishes a panel the same way).  Reference form; the device kernel k_hess_ytop computes the first sum as FP64-MFMA
// tiles.  The columns right of the panel receive Y(r, :) with everybody else's rows in the rank-2NB update.
template <class C>
KB_HD void hess_ytop_block(const C& ctx, int N, cd* W, int ld, int p0, const cd* VT, int ldvt, cd* Y, int ldy) {
    for (int idx = ctx.tid(); idx < (p0 + 1) * KB_NB; idx += ctx.nthreads()) {
        const int r = idx % (p0 + 1), t = idx / (p0 + 1);
        cd acc = czero();
        for (int c = p0 + 1; c < N; ++c) cfma(acc, W[r + (size_t)c * ld], VT[c + (size_t)t * ldvt]);
        Y[r + (size_t)t * ldy] = acc;
    }
    ctx.sync();
    for (int idx = ctx.tid(); idx < (p0 + 1) * (KB_NB - 1); idx += ctx.nthreads()) {
        const int r = idx % (p0 + 1), j = 1 + idx / (p0 + 1);
        cd acc = czero();
        for (int t = 0; t < j; ++t) acc = acc + Y[r + (size_t)t * ldy] * conj(hess_vt(W, ld, p0, p0 + j, t));
        W[r + (size_t)(p0 + j) * ld] = W[r + (size_t)(p0 + j) * ld] - acc;
    }
    ctx.sync();
}

// Extract the upper Hessenberg part of W into the work copy Hc that the QR iteration destroys
// (W itself keeps H above its Householder vectors and later feeds the inverse iteration).
template <class C>
KB_HD void hess_copy(const C& ctx, int n, const cd* W, int ld, cd* Hc, int ldc) {
    for (int idx = ctx.tid(); idx < n * n; idx += ctx.nthreads()) {
        const int i = idx % n, j = idx / n;
        Hc[i + (size_t)j * ldc] = (i <= j + 1) ? W[i + (size_t)j * ld] : czero();
    }
    ctx.sync();
}

// 2-element Householder (zlarfg with n = 2), scalar.
// On the device the square root and the three divisions are replaced by one reciprocal
// square root and one reciprocal (hardware seed + Newton steps): this routine sits on the
// serial critical path of every bulge-chasing interval.
KB_HD void larfg2(cd& alpha, cd& x, cd& tau) {
    const double xn2 = x.x * x.x + x.y * x.y;
    if (xn2 == 0.0 && alpha.y == 0.0) { tau = czero(); return; }
#if defined(__HIP_DEVICE_COMPILE__)
    const double t = fma(alpha.x, alpha.x, fma(alpha.y, alpha.y, xn2));
    double rs = __builtin_amdgcn_rsq(t);
    rs = rs * fma(-0.5 * t * rs, rs, 1.5);
    rs = rs * fma(-0.5 * t * rs, rs, 1.5);
    const double nrm = t * rs;
    const double beta = (alpha.x >= 0.0) ? -nrm : nrm;
    const double ib = (alpha.x >= 0.0) ? -rs : rs;                 // 1 / beta
    tau = mk((beta - alpha.x) * ib, -alpha.y * ib);
    // x / (alpha - beta) = x * conj(d) / |d|^2
    const double dr = alpha.x - beta, di = alpha.y;
    const double d2 = fma(dr, dr, di * di);
    double r = __builtin_amdgcn_rcp(d2);
    r = r * fma(-d2, r, 2.0);
    r = r * fma(-d2, r, 2.0);
    x = mk((x.x * dr + x.y * di) * r, (x.y * dr - x.x * di) * r);
    alpha = mk(beta, 0.0);
#else
    const double nrm = sqrt(alpha.x * alpha.x + alpha.y * alpha.y + xn2);
    const double beta = (alpha.x >= 0.0) ? -nrm : nrm;
    tau = mk((beta - alpha.x) / beta, -alpha.y / beta);
    x = cdiv(x, mk(alpha.x - beta, alpha.y));
    alpha = mk(beta, 0.0);
#endif
}

// Eigenvalues of the upper Hessenberg H (destroyed).  LAPACK zlahqr, eigenvalues only:
// every transformation touches the active block H[l..i, l..i] alone.
// All threads execute the same control flow (every decision is taken on values read
// after a barrier, or on block reductions that return identical bits to every thread).
template <class C>
KB_HD void hqr_eigvals(const C& ctx, int n, cd* H, int ld, cd* w, int* info) {
#define HH(i_, j_) H[(i_) + (size_t)(j_) * ld]
    const double ulp = KB_ULP;
    const double smlnum = KB_SAFMIN * ((double)n / ulp);
    const int tid = ctx.tid(), nt = ctx.nthreads();
    int fail = 0;
    if (n == 1) {
        if (tid == 0) { w[0] = HH(0, 0); *info = 0; }
        ctx.sync();
        return;
    }
    // make every subdiagonal real (diagonal unitary similarity; eigenvalues unchanged)
    for (int i = 1; i < n; ++i) {
        ctx.sync();
        const cd hs = HH(i, i - 1);
        if (hs.y != 0.0) {
            const double a = cabs(hs);
            const cd sc = mk(hs.x / a, -hs.y / a);   // conj(hs)/|hs|
            ctx.sync();                               // everyone has read hs
            if (tid == 0) {
                HH(i, i - 1) = mk(a, 0.0);
                if (i + 1 < n) HH(i + 1, i) = HH(i + 1, i) * conj(sc);
            }
            for (int j = i + 1 + tid; j < n; j += nt) HH(i, j) = HH(i, j) * sc;        // row i
            for (int r = tid; r < i; r += nt) HH(r, i) = HH(r, i) * conj(sc);          // column i
        }
    }
    ctx.sync();
    const int itmax = 30 * (n > 10 ? n : 10);
    int kdefl = 0;
    int i = n - 1;
    while (i >= 0) {
        int l = 0;
        bool converged = false;
        for (int its = 0; its <= itmax; ++its) {
            // ---- look for a single small subdiagonal element: largest k in (l, i]
            int kf = l;
            for (int k = l + 1 + tid; k <= i; k += nt) {
                const cd hkk1 = HH(k, k - 1);
                bool small_ = false;
                if (cabs1(hkk1) <= smlnum) small_ = true;
                else {
                    double tst = cabs1(HH(k - 1, k - 1)) + cabs1(HH(k, k));
                    if (tst == 0.0) {
                        if (k - 2 >= 0) tst += fabs(HH(k - 1, k - 2).x);
                        if (k + 1 <= n - 1) tst += fabs(HH(k + 1, k).x);
                    }
                    if (fabs(hkk1.x) <= ulp * tst) {
                        const double a1 = cabs1(hkk1), a2 = cabs1(HH(k - 1, k));
                        const double ab = fmax(a1, a2), ba = fmin(a1, a2);
                        const cd df = HH(k - 1, k - 1) - HH(k, k);
                        const double b1 = cabs1(HH(k, k)), b2 = cabs1(df);
                        const double aa = fmax(b1, b2), bb = fmin(b1, b2);
                        const double s = aa + ab;
                        if (ba * (ab / s) <= fmax(smlnum, ulp * (bb * (aa / s)))) small_ = true;
                    }
                }
                if (small_ && k > kf) kf = k;
            }
            kf = ctx.block_max(kf);
            l = kf;
            if (l > 0 && tid == 0) HH(l, l - 1) = czero();
            if (l >= i) { converged = true; break; }
            ctx.sync();
            kdefl++;
            // ---- shift
            cd t;
            if (kdefl % 20 == 0) {
                const double s = 0.75 * fabs(HH(i, i - 1).x);
                t = mk(s, 0.0) + HH(i, i);
            } else if (kdefl % 10 == 0) {
                const double s = 0.75 * fabs(HH(l + 1, l).x);
                t = mk(s, 0.0) + HH(l, l);
            } else {
                t = HH(i, i);
                const cd u = csqrt_(HH(i - 1, i)) * csqrt_(HH(i, i - 1));
                double s = cabs1(u);
                if (s != 0.0) {
                    const cd x = 0.5 * (HH(i - 1, i - 1) - t);
                    const double sx = cabs1(x);
                    s = fmax(s, sx);
                    const cd xs = mk(x.x / s, x.y / s), us = mk(u.x / s, u.y / s);
                    cd y = s * csqrt_(xs * xs + us * us);
                    if (sx > 0.0) {
                        const cd xn = mk(x.x / sx, x.y / sx);
                        if (xn.x * y.x + xn.y * y.y < 0.0) y = -y;
                    }
                    t = t - u * cdiv(u, x + y);
                }
            }
            // ---- look for two consecutive small subdiagonals: largest ms in [l+1, i-1]
            int mf = l;
            for (int mm = l + 1 + tid; mm <= i - 1; mm += nt) {
                const cd h11 = HH(mm, mm), h22 = HH(mm + 1, mm + 1);
                cd h11s = h11 - t;
                double h21 = HH(mm + 1, mm).x;
                const double s = cabs1(h11s) + fabs(h21);
                h11s = mk(h11s.x / s, h11s.y / s);
                h21 = h21 / s;
                const double h10 = HH(mm, mm - 1).x;
                if (fabs(h10) * fabs(h21) <= ulp * (cabs1(h11s) * (cabs1(h11) + cabs1(h22))))
                    if (mm > mf) mf = mm;
            }
            mf = ctx.block_max(mf);
            const int ms = mf;
            cd v1, v2;
            {
                cd h11s = HH(ms, ms) - t;
                double h21 = HH(ms + 1, ms).x;
                const double s = cabs1(h11s) + fabs(h21);
                v1 = mk(h11s.x / s, h11s.y / s);
                v2 = mk(h21 / s, 0.0);
            }
            // ---- single-shift QR sweep on rows/cols l..i
            for (int k = ms; k <= i - 1; ++k) {
                if (k > ms) { v1 = HH(k, k - 1); v2 = HH(k + 1, k - 1); }
                cd t1;
                larfg2(v1, v2, t1);
                ctx.sync();  // everyone has read H(k,k-1), H(k+1,k-1)
                if (k > ms && tid == 0) { HH(k, k - 1) = v1; HH(k + 1, k - 1) = czero(); }
                const double t2 = (t1 * v2).x;
                // rows k, k+1 ; columns k..i
                for (int j = k + tid; j <= i; j += nt) {
                    const cd a = HH(k, j), b = HH(k + 1, j);
                    const cd sum = conj(t1) * a + t2 * b;
                    HH(k, j) = a - sum;
                    HH(k + 1, j) = b - sum * v2;
                }
                ctx.sync();
                // columns k, k+1 ; rows l..min(k+2, i)
                const int rmax = (k + 2 < i) ? k + 2 : i;
                for (int j = l + tid; j <= rmax; j += nt) {
                    const cd a = HH(j, k), b = HH(j, k + 1);
                    const cd sum = t1 * a + t2 * b;
                    HH(j, k) = a - sum;
                    HH(j, k + 1) = b - sum * conj(v2);
                }
                ctx.sync();
                if (k == ms && ms > l) {
                    // the sweep started below l: make H(ms, ms-1) real again
                    cd temp = mk(1.0, 0.0) - t1;
                    const double at = cabs(temp);
                    temp = mk(temp.x / at, temp.y / at);
                    if (tid == 0) {
                        HH(ms + 1, ms) = HH(ms + 1, ms) * conj(temp);
                        if (ms + 2 <= i) HH(ms + 2, ms + 1) = HH(ms + 2, ms + 1) * temp;
                    }
                    ctx.sync();
                    // rows j in S scale (j, c>j) by temp; columns j in S scale (r<j, j) by
                    // conj(temp); S = {ms..i} \ {ms+1}.  One combined factor per element.
                    {
                        const int len = i - l + 1;
                        for (int idx = tid; idx < len * len; idx += nt) {
                            const int r = l + idx % len, c = l + idx / len;
                            if (r < c) {
                                const bool rin = (r >= ms && r != ms + 1);
                                const bool cin = (c >= ms && c != ms + 1);
                                if (rin || cin) {
                                    cd v = HH(r, c);
                                    if (rin) v = v * temp;
                                    if (cin) v = v * conj(temp);
                                    HH(r, c) = v;
                                }
                            }
                        }
                    }
                    ctx.sync();
                }
            }
            // ---- ensure H(i, i-1) is real
            {
                cd temp = HH(i, i - 1);
                ctx.sync();
                if (temp.y != 0.0) {
                    const double rt = cabs(temp);
                    if (tid == 0) HH(i, i - 1) = mk(rt, 0.0);
                    temp = mk(temp.x / rt, temp.y / rt);
                    for (int r = l + tid; r <= i - 1; r += nt) HH(r, i) = HH(r, i) * temp;
                    ctx.sync();
                }
            }
        }
        if (!converged) { fail = 1; }
        ctx.sync();
        if (tid == 0) w[i] = HH(i, i);
        if (!converged) {
            // give up on this block: report remaining diagonal entries (flagged via info)
            for (int r = l + tid; r < i; r += nt) w[r] = HH(r, r);
            i = l - 1;
        } else {
            i = l - 1;
        }
        kdefl = 0;
        ctx.sync();
    }
    if (tid == 0) *info = fail;
    ctx.sync();
#undef HH
}

// ---------------------------------------------------------------------------------
// Inverse iteration for the right eigenvectors of the upper Hessenberg H (LAPACK
// zhsein/zlaein semantics: perturb eigenvalues that coincide, replace vanishing pivots by
// eps3, start from eps3*ones, normally ONE solve reaches the growth criterion).
//
// GPU formulation - no O(n^2) factor is ever stored.  zlaein eliminates the subdiagonal of
// B = H - w I by ROW operations top-down and back-substitutes bottom-up, which forces it to
// keep U.  Here the subdiagonal is eliminated by COLUMN operations bottom-up
//      B E_{n-2} ... E_0 = R (upper triangular),  E_j mixes columns (j, j+1) with adjacent
//      column pivoting,
// so R's columns become final in the order n-1, n-2, ... - exactly the order in which the
// column-oriented back-substitution R y = b consumes them.  Elimination and substitution are
// fused into one sweep that keeps two columns and b (O(n) state, in LDS); x = E_{n-2}..E_0 y
// is a chain of n-1 2x2 updates.  Columns of H are contiguous (coalesced) and are fetched
// one step ahead of the dependent chain.  One wavefront per eigenvalue.
//   H    : n x n column-major; only rows 0..j+1 of column j are read (whatever lies below
//          the subdiagonal - e.g. Householder vectors - is ignored)
//   X    : n x n output, column k = eigenvector of w[k] in the Hessenberg basis
KB_HD int invit_scratch_bytes_per_wave(int n) { return (3 * n + 2) * (int)sizeof(cd) + ((n + 15) & ~15); }

template <class C, int MAXC>
KB_HD void invit(const C& ctx, int n, const cd* H, int ldh, const cd* w, double hnorm, cd* X, int ldx,
                 int nwaves_used, int* weak, int part = 0, int nparts = 1) {
    const int lane = ctx.lane();
    const double eps3 = fmax(hnorm * KB_ULP, KB_SAFMIN * ((double)n / KB_ULP));
    const double rootn = sqrt((double)n);
    const double growto = 0.1 / rootn;
    if (ctx.wave() >= nwaves_used) return;
    cd* cand = reinterpret_cast<cd*>(ctx.scratch()) + (size_t)ctx.wave() * (3 * n + 2);
    cd* bv = cand + n + 1;        // right-hand side, becomes y, then x
    cd* fm = bv + n + 1;          // multipliers f_j; swap flag carried in swp[]
    unsigned char* swp = reinterpret_cast<unsigned char*>(ctx.scratch()) +
                         (size_t)nwaves_used * (3 * n + 2) * sizeof(cd) + (size_t)ctx.wave() * n;
    const bool fast = (n <= MAXC * C::WS);
    int nweak = 0;
    for (int kk = part * nwaves_used + ctx.wave(); kk < n; kk += nparts * nwaves_used) {
        cd wk = w[kk];
        {
            int cnt = 0;
            for (int q = lane; q < kk; q += C::WS) cnt += (cabs1(w[q] - w[kk]) < eps3) ? 1 : 0;
            cnt = (int)ctx.wave_sum((double)cnt);
            wk.x += cnt * eps3;
        }
        bool ok = false;
        for (int its = 0; its < 4 && !ok; ++its) {
            // right-hand side (zlaein start vectors)
            for (int j = lane; j < n; j += C::WS) {
                double vj;
                if (its == 0) vj = eps3;
                else {
                    const double rtemp = eps3 / (rootn + 1.0);
                    vj = (j == 0) ? eps3 : rtemp;
                    if (j == n - its) vj -= eps3 * rootn;
                }
                bv[j] = mk(vj, 0.0);
            }
            // candidate for position n-1: raw column n-1 of B (rows 0..n-1)
            for (int r = lane; r < n; r += C::WS) {
                cd v = H[r + (size_t)(n - 1) * ldh];
                if (r == n - 1) v = v - wk;
                cand[r] = v;
            }
            ctx.wave_fence();
            bool rescaled = false;
            // prefetch registers for the raw column of the next step (rows lane + WS*c)
            cd nxt[MAXC];
            if (fast && n >= 2) {
#pragma unroll
                for (int c = 0; c < MAXC; ++c) {
                    const int r = lane + c * C::WS;
                    nxt[c] = (r <= n - 1) ? H[r + (size_t)(n - 2) * ldh] : czero();
                }
            }
            for (int j = n - 2; j >= 0; --j) {
                // raw column j of B, rows 0..j+1
                cd cur[MAXC];
                if (fast) {
#pragma unroll
                    for (int c = 0; c < MAXC; ++c) cur[c] = nxt[c];
                    if (j >= 1) {
#pragma unroll
                        for (int c = 0; c < MAXC; ++c) {
                            const int r = lane + c * C::WS;
                            nxt[c] = (r <= j) ? H[r + (size_t)(j - 1) * ldh] : czero();
                        }
                    }
                }
                const cd rsub = H[(j + 1) + (size_t)j * ldh];      // B(j+1, j) = H(j+1, j)
                cd cpiv = cand[j + 1];                              // diagonal of the candidate
                cd f, piv;
                const bool swap = cabs1(cpiv) < cabs1(rsub);
                if (!swap) {
                    if (is_zero(cpiv)) cpiv = mk(eps3, 0.0);
                    piv = cpiv;
                    f = cdiv(rsub, cpiv);
                } else {
                    piv = rsub;
                    f = cdiv(cpiv, rsub);
                }
                // y_{j+1} and the right-hand side update use the FINAL column at position j+1
                const cd yj1 = cdiv(bv[j + 1], piv);
                ctx.wave_fence();
                if (fast) {
#pragma unroll
                    for (int c = 0; c < MAXC; ++c) {
                        const int r = lane + c * C::WS;
                        if (r <= j) {
                            cd raw = cur[c];
                            if (r == j) raw = raw - wk;
                            const cd cn = cand[r];
                            const cd fin = swap ? raw : cn;
                            const cd oth = swap ? cn : raw;
                            cand[r] = oth - f * fin;          // new candidate for position j
                            bv[r] = bv[r] - yj1 * fin;
                        }
                    }
                } else {
                    for (int r = lane; r <= j; r += C::WS) {
                        cd raw = H[r + (size_t)j * ldh];
                        if (r == j) raw = raw - wk;
                        const cd cn = cand[r];
                        const cd fin = swap ? raw : cn;
                        const cd oth = swap ? cn : raw;
                        cand[r] = oth - f * fin;
                        bv[r] = bv[r] - yj1 * fin;
                    }
                }
                if (lane == 0) {
                    bv[j + 1] = yj1;
                    fm[j] = f;
                    swp[j] = swap ? 1 : 0;
                }
                // guard against overflow: rescale the whole system
                if (cabs1(yj1) > 1e120) {
                    ctx.wave_fence();
                    for (int r = lane; r < n; r += C::WS) bv[r] = bv[r] * 1e-120;
                    rescaled = true;
                }
                ctx.wave_fence();
            }
            {
                cd p0 = cand[0];
                if (is_zero(p0)) p0 = mk(eps3, 0.0);
                const cd y0 = cdiv(bv[0], p0);
                ctx.wave_fence();
                if (lane == 0) {
                    // x = E_{n-2} ... E_0 y : E_j acts on coordinates (j, j+1); only one value
                    // is carried, the loads of y_{j+1}, f_j do not depend on it
                    cd carry = y0;
                    for (int j = 0; j <= n - 2; ++j) {
                        const cd b = bv[j + 1], f = fm[j];
                        const cd t = b - f * carry;
                        if (!swp[j]) { bv[j] = carry; carry = t; }
                        else { bv[j] = t; }
                    }
                    bv[n - 1] = carry;
                }
                ctx.wave_fence();
            }
            double vn = 0.0;
            for (int j = lane; j < n; j += C::WS) vn += cabs1(bv[j]);
            vn = ctx.wave_sum(vn);
            ok = rescaled || (vn >= growto);
        }
        if (!ok) nweak++;
        double mx = 0.0;
        for (int j = lane; j < n; j += C::WS) mx = fmax(mx, cabs1(bv[j]));
        mx = ctx.wave_max(mx);
        const double inv = (mx > 0.0) ? 1.0 / mx : 1.0;
        for (int j = lane; j < n; j += C::WS) X[j + (size_t)kk * ldx] = bv[j] * inv;
        ctx.wave_fence();
    }
    if (lane == 0 && nweak > 0) *weak = 1;   // benign race: every writer stores 1
}

#if defined(__HIPCC__)
// Register-resident inverse iteration (device only): the same elimination and the same arithmetic as
// `invit` above, but the three O(n) vectors of a solve (candidate column, right-hand side, multipliers) live
// in the REGISTERS of the wavefront - element r on lane r mod 64, chunk r / 64 - instead of LDS: a value at a
// known position is broadcast with v_readlane, no fences (the LDS form holds 19 KB per solve at n = 400).  Only the
// multipliers f_j, written once per step by one lane and read back by the serial substitution chain, sit in LDS
// (fmL: MAXC * 64 entries per wavefront).
// The chunk of the pivot position is a compile-time index (outer loops over chunks are unrolled), so every
// register access is static.  n <= MAXC * 64.
__device__ __forceinline__ cd kb_bcast(cd v, int l) {
    return mk(__hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v.x), l), __builtin_amdgcn_readlane(__double2loint(v.x), l)),
              __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v.y), l), __builtin_amdgcn_readlane(__double2loint(v.y), l)));
}

// One elimination step of invit_reg (position p: column j = p - 1 is eliminated against the candidate at p).  The chunk
// CJ of row p is a compile-time index; LP0 marks the one step per chunk whose row j lies in the chunk below (p = 64 CJ).
// What is wavefront-uniform is decided by scalar branches (the pivot choice) or known at compile time (which chunks are
// full, where the diagonal / the new y / the multiplier go), so the vector part of a chunk is eight FMAs.
__device__ __forceinline__ cd csel2(bool c, cd a, cd b) { return mk(c ? a.x : b.x, c ? a.y : b.y); }

// STREAM (members of more than 512 rows): the raw column is not double-buffered in registers - only the chunk that holds
// the pivot row is fetched one step ahead (nxt[0]), the others are loaded where they are used - and the multipliers go to
// global memory (the output column of this eigenvalue, free until the solve is over) instead of LDS: 2 x MAXC chunks of
// registers instead of 4, so that 20 chunks (1280 rows) fit.
constexpr int KB_INVIT_BIG_MAXC = 20;     // register chunks of the streaming form: members of up to 1280 rows

// value of lane `src` (any lane, 0..63; out-of-range sources are masked by the caller) through the LDS crossbar
__device__ __forceinline__ cd kb_lane_from(cd v, int src) {
    const int a = (src & 63) << 2;
    return mk(__hiloint2double(__builtin_amdgcn_ds_bpermute(a, __double2hiint(v.x)), __builtin_amdgcn_ds_bpermute(a, __double2loint(v.x))),
              __hiloint2double(__builtin_amdgcn_ds_bpermute(a, __double2hiint(v.y)), __builtin_amdgcn_ds_bpermute(a, __double2loint(v.y))));
}

template <int MAXC, bool STREAM = false>
struct InvitRegState {
    cd cand[MAXC], bv[MAXC], nxt[STREAM ? 1 : MAXC];
    cd* fmL;                                                  // multipliers f_j of this solve: LDS, MAXC * 64 entries per wavefront (STREAM: global)
    unsigned swp;
    bool rescaled;
    cd wk;
    double eps3;
    const cd* __restrict__ H;
    int ldh, lane;

    template <int CJ, bool LP0>
    __device__ __forceinline__ void step(int p) {
        constexpr int CK = LP0 ? CJ - 1 : CJ;                 // chunk of row j
        const int j = p - 1;
        const int lp = LP0 ? 0 : p - 64 * CJ;                 // lane of row p (chunk CJ)
        const int lj = LP0 ? 63 : lp - 1;                     // lane of row j (chunk CK)
        cd cur[MAXC];
        if constexpr (!STREAM) {
#pragma unroll
            for (int c = 0; c <= CJ; ++c) cur[c] = nxt[c];
            if (j >= 1) {                                      // raw column j - 1, rows <= j, for the next step
                const cd* __restrict__ Hc = H + (size_t)(j - 1) * ldh + lane;
#pragma unroll
                for (int c = 0; c < CK; ++c) nxt[c] = Hc[64 * c];
                nxt[CK] = (LP0 || lane <= lj) ? Hc[64 * CK] : czero();
            }
        } else {
            cur[CJ] = nxt[0];                                  // the chunk of row p, fetched during the step before
            if (j >= 1) {                                      // ... and the one of row p - 1 of column j - 1 for the next step
                constexpr int CN = LP0 ? CJ - 1 : CJ;
                nxt[0] = (lane + 64 * CN <= p - 1) ? H[(size_t)(j - 1) * ldh + lane + 64 * CN] : czero();
            }
            if constexpr (LP0) cur[CK] = H[(size_t)j * ldh + lane + 64 * CK];     // (row j sits in the chunk below)
        }
        // STREAM: the chunks below CK are loaded where they are used, four at a time (a compiler barrier keeps the loads of
        // the next group from being hoisted: with all of a column in flight the allocation spills)
        const cd* __restrict__ Hj = H + (size_t)j * ldh + lane;
        (void)Hj;
        const cd rsub = kb_bcast(cur[CJ], lp);                // B(j+1, j) = H(j+1, j)
        cd cpiv = kb_bcast(cand[CJ], lp);                     // diagonal of the candidate
        const cd bp = kb_bcast(bv[CJ], lp);
        // one complex reciprocal (v_rcp_f64 + two Newton steps on |piv|^2) and two products instead of two Smith
        // divisions: the pivots are O(||H||) .. eps3 = O(ulp ||H||), far from the range limits of |.|^2
        const bool swap = __builtin_amdgcn_readfirstlane((int)(cabs1(cpiv) < cabs1(rsub))) != 0;
        if (!swap && is_zero(cpiv)) cpiv = mk(eps3, 0.0);
        const cd piv = swap ? rsub : cpiv;
        const cd oth = swap ? cpiv : rsub;
        const double d2 = fma(piv.x, piv.x, piv.y * piv.y);
        double rr = __builtin_amdgcn_rcp(d2);
        rr = rr * fma(-d2, rr, 2.0);
        rr = rr * fma(-d2, rr, 2.0);
        const cd pinv = mk(piv.x * rr, -piv.y * rr);
        const cd f = oth * pinv;
        const cd yj1 = bp * pinv;
        {                                                      // the diagonal entry of the raw column: B(j, j) = H(j, j) - w
            const bool dj = lane == lj;
            cur[CK] = mk(dj ? cur[CK].x - wk.x : cur[CK].x, dj ? cur[CK].y - wk.y : cur[CK].y);
        }
        const bool top = LP0 || lane <= lj;                    // rows <= j of chunk CK
        if constexpr (STREAM) {
            // the chunks below CK in groups of four: the group's loads, then ONE scalar branch on the pivot choice.  The
            // compiler barrier in front of every group keeps later groups' loads (which both sides of the branch share, so
            // that they would be hoisted above it) from being issued early: with a whole column in flight the allocation
            // spills by the kilobyte.
#pragma unroll
            for (int c0 = 0; c0 < CK; c0 += 4) {
                asm volatile("" ::: "memory");
                cd raw[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (c0 + q < CK) raw[q] = Hj[64 * (c0 + q)];
                if (swap) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (c0 + q < CK) {
                            cand[c0 + q] = cand[c0 + q] - f * raw[q];
                            bv[c0 + q] = bv[c0 + q] - yj1 * raw[q];
                        }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (c0 + q < CK) {
                            const cd cn = cand[c0 + q];
                            cand[c0 + q] = raw[q] - f * cn;
                            bv[c0 + q] = bv[c0 + q] - yj1 * cn;
                        }
                }
            }
            asm volatile("" ::: "memory");
            const cd raw = cur[CK], cn = cand[CK];
            const cd nc = swap ? cn - f * raw : raw - f * cn;
            const cd nb = bv[CK] - yj1 * (swap ? raw : cn);
            cand[CK] = csel2(top, nc, cn);
            bv[CK] = csel2(top, nb, bv[CK]);
        } else if (swap) {                                     // the raw column is final at position j + 1
#pragma unroll
            for (int c = 0; c < CK; ++c) {
                const cd raw = cur[c];
                cand[c] = cand[c] - f * raw;
                bv[c] = bv[c] - yj1 * raw;
            }
            const cd raw = cur[CK];
            const cd nc = cand[CK] - f * raw, nb = bv[CK] - yj1 * raw;
            cand[CK] = csel2(top, nc, cand[CK]);
            bv[CK] = csel2(top, nb, bv[CK]);
        } else {                                               // the candidate is
#pragma unroll
            for (int c = 0; c < CK; ++c) {
                const cd cn = cand[c];
                cand[c] = cur[c] - f * cn;
                bv[c] = bv[c] - yj1 * cn;
            }
            const cd cn = cand[CK];
            const cd nc = cur[CK] - f * cn, nb = bv[CK] - yj1 * cn;
            cand[CK] = csel2(top, nc, cn);
            bv[CK] = csel2(top, nb, bv[CK]);
        }
        bv[CJ] = csel2(lane == lp, yj1, bv[CJ]);              // y at row p
        if (lane == 0) fmL[j] = f;
        if (lane == lj) swp = swap ? (swp | (1u << CK)) : (swp & ~(1u << CK));
        if (cabs1(yj1) > 1e120) {                              // guard against overflow: rescale the whole system
#pragma unroll
            for (int c = 0; c < MAXC; ++c) bv[c] = bv[c] * 1e-120;
            rescaled = true;
        }
    }

    // all positions of chunk CJ, top down
    template <int CJ>
    __device__ __forceinline__ void chunk(int n) {
        if constexpr (STREAM) {
            // the first chunk of the solve: fetch the pivot-row chunk of column n - 2 (rows <= n - 1)
            if (n - 1 < 64 * (CJ + 1)) nxt[0] = (n >= 2 && lane + 64 * CJ <= n - 1) ? H[(size_t)(n - 2) * ldh + lane + 64 * CJ] : czero();
        }
        const int phi = (n - 1 < 64 * CJ + 63) ? n - 1 : 64 * CJ + 63;
        const int plo = 64 * CJ + 1;                           // lp >= 1
        for (int p = phi; p >= plo; --p) step<CJ, false>(p);
        if constexpr (CJ >= 1) {
            if (n - 1 >= 64 * CJ) step<CJ, true>(64 * CJ);
        }
    }
};

template <int MAXC, bool STREAM, int CJ>
__device__ __forceinline__ void invit_chunks(InvitRegState<MAXC, STREAM>& S, int n) {
    if (n - 1 >= 64 * CJ) S.template chunk<CJ>(n);
    if constexpr (CJ >= 1) invit_chunks<MAXC, STREAM, CJ - 1>(S, n);
}

template <int MAXC, bool STREAM = false>
__device__ void invit_reg(const DevCtx& ctx, int n, const cd* __restrict__ H, int ldh, const cd* __restrict__ w,
                          double hnorm, cd* __restrict__ X, int ldx, int kk_begin, int kk_step, int* weak, cd* fmL) {
    const int lane = ctx.lane();
    const double eps3 = fmax(hnorm * KB_ULP, KB_SAFMIN * ((double)n / KB_ULP));
    const double rootn = sqrt((double)n);
    const double growto = 0.1 / rootn;
    int nweak = 0;
    for (int kk = kk_begin; kk < n; kk += kk_step) {
        cd wk = w[kk];
        {
            int cnt = 0;
            for (int q = lane; q < kk; q += 64) cnt += (cabs1(w[q] - w[kk]) < eps3) ? 1 : 0;
            cnt = (int)ctx.wave_sum((double)cnt);
            wk.x += cnt * eps3;
        }
        InvitRegState<MAXC, STREAM> S;
        if (STREAM) fmL = X + (size_t)kk * ldx;          // the multipliers borrow the output column
        cd (&cand)[MAXC] = S.cand;
        cd (&bv)[MAXC] = S.bv;
        unsigned& swp = S.swp;             // bit c: swap flag of element lane + 64 c
        S.wk = wk; S.eps3 = eps3; S.H = H; S.ldh = ldh; S.lane = lane; S.fmL = fmL;
        bool ok = false;
        for (int its = 0; its < 4 && !ok; ++its) {
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const int r = lane + 64 * c;
                double vj;
                if (its == 0) vj = eps3;
                else {
                    const double rtemp = eps3 / (rootn + 1.0);
                    vj = (r == 0) ? eps3 : rtemp;
                    if (r == n - its) vj -= eps3 * rootn;
                }
                bv[c] = mk(vj, 0.0);
                cd v = (r < n) ? H[r + (size_t)(n - 1) * ldh] : czero();
                if (r == n - 1) v = v - wk;
                cand[c] = v;
            }
            swp = 0;
            S.rescaled = false;
            if constexpr (!STREAM) {
#pragma unroll
                for (int c = 0; c < MAXC; ++c) {
                    const int r = lane + 64 * c;
                    S.nxt[c] = (n >= 2 && r <= n - 1) ? H[r + (size_t)(n - 2) * ldh] : czero();
                }
            }
            // position p = j + 1 runs from n-1 down to 1; the chunk of p is a compile-time index
            invit_chunks<MAXC, STREAM, MAXC - 1>(S, n);
            const bool rescaled = S.rescaled;
            {
                cd p0 = kb_bcast(cand[0], 0);
                if (is_zero(p0)) p0 = mk(eps3, 0.0);
                const cd y0 = cdiv(kb_bcast(bv[0], 0), p0);
                // x = E_{n-2} ... E_0 y : E_j acts on coordinates (j, j+1); one value is carried:
                //     sw_j:  x_j = y_{j+1} - f_j c_j,  c_{j+1} = c_j          else:  x_j = c_j,  c_{j+1} = y_{j+1} - f_j c_j
                // The carry obeys an affine recurrence c_{j+1} = a_j c_j + b_j, (a_j, b_j) = sw_j ? (1, 0) : (-f_j, y_{j+1}):
                // per chunk of 64 steps the maps are composed by a lane-parallel prefix scan (six rounds through the LDS
                // crossbar) instead of 64 serial steps of broadcasts - the chain was more than a quarter of a solve.
                cd carry = y0;
#pragma unroll
                for (int cj = 0; cj < MAXC; ++cj) {
                    const int jhi = (n - 2 < 64 * cj + 63) ? n - 2 : 64 * cj + 63;
                    if (64 * cj > jhi) continue;
                    const bool valid = lane + 64 * cj <= jhi;
                    const cd f = valid ? fmL[lane + 64 * cj] : czero();
                    const bool sw = ((swp >> cj) & 1u) != 0;
                    // y_{j+1}: the next lane's entry, lane 63 takes lane 0 of the next chunk
                    const cd up = kb_lane_from(bv[cj], (lane + 1) & 63);
                    const cd nx = kb_bcast(bv[(cj + 1 < MAXC) ? cj + 1 : cj], 0);
                    const cd yn = csel2(lane == 63, nx, up);
                    const bool idm = sw || !valid;                      // identity map (a swap step or beyond the chain)
                    cd A = idm ? mk(1.0, 0.0) : -f;
                    cd B = idm ? czero() : yn;
#pragma unroll
                    for (int dsh = 1; dsh < 64; dsh <<= 1) {            // inclusive scan: (A, B)_j <- (A, B)_j o (A, B)_{j - dsh}
                        const cd Ap = kb_lane_from(A, lane - dsh), Bp = kb_lane_from(B, lane - dsh);
                        const bool on = lane >= dsh;
                        const cd nB = A * Bp + B, nA = A * Ap;
                        B = csel2(on, nB, B);
                        A = csel2(on, nA, A);
                    }
                    // the carry entering step j: the composed map of the steps before it applied to the chunk's carry
                    const cd Ae = kb_lane_from(A, lane - 1), Be = kb_lane_from(B, lane - 1);
                    const cd cin = (lane == 0) ? carry : Ae * carry + Be;
                    const cd xj = sw ? yn - f * cin : cin;
                    if (valid) bv[cj] = xj;
                    carry = kb_bcast(A, 63) * carry + kb_bcast(B, 63);  // lanes beyond the chain hold identity maps
                }
#pragma unroll
                for (int c = 0; c < MAXC; ++c)
                    if (lane + 64 * c == n - 1) bv[c] = carry;
            }
            double vn = 0.0;
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (lane + 64 * c < n) vn += cabs1(bv[c]);
            vn = ctx.wave_sum(vn);
            ok = rescaled || (vn >= growto);
        }
        if (!ok) nweak++;
        double mx = 0.0;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (lane + 64 * c < n) mx = fmax(mx, cabs1(bv[c]));
        mx = ctx.wave_max(mx);
        const double inv = (mx > 0.0) ? 1.0 / mx : 1.0;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int r = lane + 64 * c;
            if (r < n) X[r + (size_t)kk * ldx] = bv[c] * inv;
        }
    }
    if (lane == 0 && nweak > 0) *weak = 1;   // benign race: every writer stores 1
}
#endif

}  // namespace kb
