// Bidiagonal SVD by divide and conquer: B (upper bidiagonal, real d / e of the Householder reduction) = X diag(s) Y^T.
//
// Replaces the QR-iteration route (logged rotations replayed on Q and P: 62.7 of the SVD's 84 m^3 flops as plane
// rotations) with what LAPACK's zgesdd - the driver behind the reference's scipy.linalg.svd, kbdm.py:166 - does:
// dbdsdc.  Own derivation in the organisation of dlasd0-4 (tools/proto_bdsdc.py is the numpy statement of it):
//
//   * tree: node = rows [lo, lo + n) of B with columns [lo, lo + n + sqre); a node of more than KB_DC_LEAF rows splits
//     into  left (nl = (n-1)/2 rows, sqre 1) | centre row | right (n - nl - 1 rows, the node's sqre).  All members of a
//     depth are solved together; depth dc_depth(m) holds the leaves.
//   * leaf: one-sided Jacobi on the columns of the (n x n+sqre) block: V = the accumulated rotations (the column that
//     ends smallest is the null vector when sqre = 1), s = column norms, U = normalised columns.
//   * merge: in the children's bases the node is  M = [z^T; 0 D]  (d_0 = 0 pairs with the centre row); singular values
//     = roots of the secular equation 1/rho + sum z_j^2 / (d_j^2 - x) = 0 (x = sigma^2; safeguarded rational
//     iteration around the nearer pole, sigma kept as pole + tau so that every difference d_j - sigma is exact to
//     working precision), z recomputed from the roots (Loewner / Gu-Eisenstat) so that the vectors
//     v_i = (z_j / (d_j^2 - sigma_i^2))_j,  u_i = (-1, d_j v_i(j))  are orthogonal to working precision whatever the
//     roots' last bits.  Deflation: negligible z_j, and close d's rotated into one.
//   * ONE dense coefficient matrix per side and node: permutation to sorted order, deflation rotations, deflated unit
//     columns and the null-column rotation are folded into it, so the node's vectors are plain products
//     U = Ubasis CU,  V = Vbasis CV  with the block-diagonal bases [U1 . .; . 1 .; . . U2] and [V1 .; . V2]  - real
//     GEMMs (FP64 MFMA tiles on the device), 4/3 * 4 m^3 flops over all levels.
//
// Per-member workspace (doubles, column-major m x m arrays with ld = m): U[2], V[2] (vectors of alternate depths, node
// blocks on the diagonal), CU, CV (coefficients), D[2] (singular values of alternate depths, a node's at [lo, lo + n)).
// Non-root nodes deliver their singular values ascending (null column of V last); the root descending.
#pragma once
#include "kb_ctx.hpp"

namespace kb {

constexpr int KB_DC_LEAF = 32;          // largest leaf (rows)
constexpr int KB_DC_MAXIT = 60;         // secular iterations per root (the safeguard bisects: it always ends)

struct DcNode { int lo, n, sqre; };

KB_HD int dc_depth(int m) {
    int L = 0, s = m;
    while (s > KB_DC_LEAF) { s = s - 1 - (s - 1) / 2; ++L; }      // the larger child
    return L;
}
KB_HD DcNode dc_node(int m, int depth, int idx) {
    DcNode nd; nd.lo = 0; nd.n = m; nd.sqre = 0;
    for (int b = depth - 1; b >= 0; --b) {
        const int nl = (nd.n - 1) / 2;
        if ((idx >> b) & 1) { nd.lo += nl + 1; nd.n = nd.n - nl - 1; }
        else { nd.n = nl; nd.sqre = 1; }
    }
    return nd;
}

struct DcWs {
    int m;
    double* U[2]; double* V[2]; double* CU; double* CV; double* D[2];
};
KB_HD long long dc_ws_doubles(int m) { return 6LL * m * m + 2LL * m + 16; }
KB_HD DcWs dc_ws(double* base, int m) {
    DcWs w; w.m = m;
    const size_t M = (size_t)m * m;
    w.U[0] = base; w.U[1] = base + M; w.V[0] = base + 2 * M; w.V[1] = base + 3 * M;
    w.CU = base + 4 * M; w.CV = base + 5 * M; w.D[0] = base + 6 * M; w.D[1] = base + 6 * M + m;
    return w;
}

// ------------------------------------------------------------------------------------------------ leaf
KB_HD int dc_leaf_scratch_bytes(int n) {
    const int cm = n + 1, ldw = n | 1, ldv = cm | 1;
    return (int)sizeof(double) * (ldw * cm + ldv * cm + 2 * cm) + (int)sizeof(int) * (2 * cm + 8) + 64;
}

// Scale of a member's bidiagonal: every kernel of the divide and conquer works on (d, e) / dc_scale and the root
// multiplies the singular values back (LAPACK dbdsdc scales by the max-norm likewise).  All threads must call.
template <class C>
KB_HD double dc_scale(const C& ctx, const double* d, const double* e, int m) {
    double mx = 0.0;
    for (int i = ctx.tid(); i < m; i += ctx.nthreads()) mx = fmax(mx, fmax(fabs(d[i]), (i + 1 < m) ? fabs(e[i]) : 0.0));
    mx = ctx.block_max(mx);
    return (mx > 0.0 && mx < 1e300) ? mx : 1.0;
}

// SVD of the leaf block rows [lo, lo+n) x columns [lo, lo+n+sqre) of B / scale.  One workgroup (one wavefront is enough).
// Columns whose norm falls below eps * ||block||_F are numerically zero: they leave the iteration (their norm would go
// on shrinking quadratically without ever meeting the relative criterion), keep their norm as singular value and get
// their left vector from an orthonormal completion - an error of eps * ||block|| in the factorisation, the accuracy
// the divide and conquer has anyway.
template <class C>
KB_HD void dc_leaf(const C& ctx, const double* d, const double* e, int m, DcNode nd, double* Udst, double* Vdst,
                   double* Ddst, bool descending, double scale) {
    const int n = nd.n, c = n + nd.sqre, lo = nd.lo;
    const int cm = n + 1, ldw = n | 1, ldv = cm | 1;
    double* W = reinterpret_cast<double*>(ctx.scratch());
    double* V = W + ldw * cm;
    double* nrm = V + ldv * cm;           // norms (cm), then a work vector (cm)
    double* wk = nrm + cm;
    int* ord = reinterpret_cast<int*>(nrm + 2 * cm);
    int* pick = ord + 2 * cm;
    const int T = ctx.nthreads(), t = ctx.tid();
    const double sinv = 1.0 / scale;
    for (int x = t; x < ldw * c; x += T) W[x] = 0.0;
    for (int x = t; x < ldv * c; x += T) V[x] = 0.0;
    ctx.sync();
    double fro = 0.0;
    for (int r = t; r < n; r += T) {
        const double dv = d[lo + r] * sinv;
        W[r + r * ldw] = dv; fro = fma(dv, dv, fro);
        if (r + 1 < c) { const double ev = e[lo + r] * sinv; W[r + (r + 1) * ldw] = ev; fro = fma(ev, ev, fro); }
    }
    for (int r = t; r < c; r += T) V[r + r * ldv] = 1.0;
    fro = ctx.block_sum(fro);
    const double thr2 = (KB_EPS * KB_EPS) * fro;          // squared norm of a negligible column
    const int ce = (c + 1) & ~1;               // players of the round-robin (an extra dummy when c is odd)
    // A pair of columns is rotated by C::QUAD adjacent lanes (device: four, each a quarter of the rows; the dot products
    // meet in a DPP butterfly, so the four lanes decide on identical bits), the pairs of a round by different lane groups
    const int LP = C::QUAD, sub = t % LP, slot = t / LP, nslot = (T / LP) > 0 ? T / LP : 1;
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotated = 0;
        for (int rd = 0; rd < ce - 1; ++rd) {
            for (int pr0 = 0; pr0 < ce / 2; pr0 += nslot) {               // (uniform trip count: the butterfly needs every lane)
                const int pr = pr0 + slot;
                bool act = pr < ce / 2;
                int p = 0, q = 0;
                if (act) {
                    if (pr == 0) { p = ce - 1; q = rd; }
                    else { p = (rd + pr) % (ce - 1); q = (rd - pr + (ce - 1)) % (ce - 1); }
                    if (p > q) { const int x = p; p = q; q = x; }
                    if (q >= c) act = false;
                }
                double* wp = W + p * ldw; double* wq = W + q * ldw;
                double a = 0.0, b = 0.0, g = 0.0;
                if (act)
                    for (int r = sub; r < n; r += LP) { a = fma(wp[r], wp[r], a); b = fma(wq[r], wq[r], b); g = fma(wp[r], wq[r], g); }
                a = ctx.quad_sum(a); b = ctx.quad_sum(b); g = ctx.quad_sum(g);
                if (!act || a <= thr2 || b <= thr2) continue;
                if (g == 0.0 || fabs(g) <= KB_EPS * sqrt(a * b)) continue;
                rotated = 1;
                const double zeta = (b - a) / (2.0 * g);
                const double tt = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
                for (int r = sub; r < n; r += LP) {
                    const double x = wp[r], y = wq[r];
                    wp[r] = cs * x - sn * y; wq[r] = sn * x + cs * y;
                }
                double* vp = V + p * ldv; double* vq = V + q * ldv;
                for (int r = sub; r < c; r += LP) {
                    const double x = vp[r], y = vq[r];
                    vp[r] = cs * x - sn * y; vq[r] = sn * x + cs * y;
                }
            }
            ctx.sync();
        }
        if (ctx.block_max(rotated) == 0) break;
    }
    for (int j = t; j < c; j += T) {
        double a = 0.0;
        for (int r = 0; r < n; ++r) a = fma(W[r + j * ldw], W[r + j * ldw], a);
        nrm[j] = sqrt(a);
    }
    ctx.sync();
    // rank by norm ascending (ties by index): with sqre the smallest is the null column and goes last
    for (int j = t; j < c; j += T) {
        int r = 0;
        for (int k = 0; k < c; ++k) r += (nrm[k] < nrm[j] || (nrm[k] == nrm[j] && k < j)) ? 1 : 0;
        int pos;
        if (nd.sqre) pos = (r == 0) ? n : r - 1; else pos = r;           // ascending position among the n values
        if (descending && pos < n) pos = n - 1 - pos;
        ord[pos] = j;
    }
    ctx.sync();
    const double thr = sqrt(thr2);
    for (int k = 0; k < n; ++k) {
        const int j = ord[k];
        const double s = nrm[j];
        const double inv = (s > thr) ? 1.0 / s : 0.0;          // numerically zero columns: completed below
        for (int r = t; r < n; r += T) Udst[(lo + r) + (size_t)(lo + k) * m] = W[r + j * ldw] * inv;
        if (t == 0) Ddst[lo + k] = s;
    }
    for (int k = 0; k < c; ++k) {
        const int j = ord[k];
        for (int r = t; r < c; r += T) Vdst[(lo + r) + (size_t)(lo + k) * m] = V[r + j * ldv];
    }
    ctx.sync();
    // orthonormal completion of U for the numerically zero columns, one after the other: the unit vector with the
    // largest component outside the span of the columns so far, orthogonalised twice
    double* Ub = Udst + lo + (size_t)lo * m;
    for (int k = 0; k < n; ++k) {
        if (nrm[ord[k]] > thr) continue;                       // (uniform: nrm and ord are shared)
        double best = -1.0; int bi = 0;
        for (int r = t; r < n; r += T) {
            double rem = 1.0;
            for (int j = 0; j < n; ++j) {
                if (j == k || (nrm[ord[j]] <= thr && j > k)) continue;
                const double u = Ub[r + (size_t)j * m];
                rem -= u * u;
            }
            if (rem > best) { best = rem; bi = r; }
        }
        const double bmax = ctx.block_max(best);
        if (best == bmax) pick[0] = bi;                        // any of the maximisers
        ctx.sync();
        const int cand = pick[0];
        for (int r = t; r < n; r += T) {
            double u = (r == cand) ? 1.0 : 0.0;
            for (int j = 0; j < n; ++j) {
                if (j == k || (nrm[ord[j]] <= thr && j > k)) continue;
                u -= Ub[cand + (size_t)j * m] * Ub[r + (size_t)j * m];
            }
            wk[r] = u;
        }
        ctx.sync();
        for (int j = t; j < n; j += T) {                       // second pass: coefficients, then the update
            double dot = 0.0;
            if (!(j == k || (nrm[ord[j]] <= thr && j > k)))
                for (int r = 0; r < n; ++r) dot += Ub[r + (size_t)j * m] * wk[r];
            W[j] = dot;                                        // (W is free now)
        }
        ctx.sync();
        double nn = 0.0;
        for (int r = t; r < n; r += T) {
            double u = wk[r];
            for (int j = 0; j < n; ++j) u -= W[j] * Ub[r + (size_t)j * m];
            wk[r] = u; nn = fma(u, u, nn);
        }
        nn = ctx.block_sum(nn);
        const double inv = 1.0 / sqrt(nn);
        for (int r = t; r < n; r += T) Ub[r + (size_t)k * m] = wk[r] * inv;
        ctx.sync();
    }
}

// ------------------------------------------------------------------------------------------------ secular equation
// eta in (a, b) with  c + S / (a - eta) + R / (b - eta) = 0   (a < 0 < b, S, R >= 0); false if there is none
KB_HD bool dc_quad(double c, double a, double S, double b, double R, double& eta) {
    const double qa = c, qb = -(c * (a + b) + S + R), qc = c * a * b + S * b + R * a;
    double r1, r2;
    if (qa == 0.0) {
        if (qb == 0.0) return false;
        r1 = r2 = -qc / qb;
    } else {
        const double disc = qb * qb - 4.0 * qa * qc;
        if (!(disc >= 0.0)) return false;
        const double q = -0.5 * (qb + (qb >= 0.0 ? sqrt(disc) : -sqrt(disc)));
        r1 = q / qa;
        r2 = (q != 0.0) ? qc / q : r1;
    }
    const bool ok1 = (r1 > a && r1 < b), ok2 = (r2 > a && r2 < b);
    if (ok1 && ok2) eta = (fabs(r1) < fabs(r2)) ? r1 : r2;
    else if (ok1) eta = r1;
    else if (ok2) eta = r2;
    else return false;
    return true;
}

// Root i of g(x) = rho_inv + sum_j z2_j / (dk_j^2 - x) in (dk_i^2, dk_{i+1}^2) (the last one beyond dk_{K-1}^2).
// Returns the origin pole o and tau: sigma_i = dk[o] + tau.  Returns false if the iteration limit was hit.
KB_HD bool dc_secular(int i, int K, const double* dk, const double* z2, double rho_inv, int& o_out, double& tau_out) {
    const bool last = (i == K - 1);
    int o;
    double lo, hi, tau;
    if (K == 1) { o_out = 0; tau_out = sqrt(1.0 / rho_inv); return true; }
    if (!last) {
        const double dl = dk[i], dr = dk[i + 1];
        const double mid = 0.5 * (dr - dl);
        // g at the midpoint: the two neighbouring poles apart from the rest
        double rest = rho_inv;
        for (int j = 0; j < K; ++j) {
            if (j == i || j == i + 1) continue;
            const double del = (dk[j] - dl) - mid, w = (dk[j] + dl) + mid;
            rest += z2[j] / (del * w);
        }
        const double a = -mid * ((dl + dl) + mid);             // dl^2 - x_mid
        const double b = mid * ((dr + dl) + mid);              // dr^2 - x_mid   ((dr - dl) - mid = mid)
        const double gm = rest + z2[i] / a + z2[i + 1] / b;
        const double sigm = dl + mid;
        double eta;
        if (gm > 0.0) { o = i; lo = 0.0; hi = mid; tau = mid; }
        else { o = i + 1; lo = -mid; hi = 0.0; tau = -mid; }
        if (dc_quad(rest, a, z2[i], b, z2[i + 1], eta)) {       // first guess: the two-pole model at the midpoint
            const double x2 = sigm * sigm + eta;
            if (x2 > 0.0) {
                const double sg = sigm + eta / (sigm + sqrt(x2));
                const double cand = (o == i) ? sg - dl : sg - dr;
                if (cand > lo && cand < hi) tau = cand;
            }
        }
        if (tau == lo || tau == hi) tau = 0.5 * (lo + hi);
    } else {
        o = K - 1;
        const double rho = 1.0 / rho_inv;
        hi = rho / (dk[o] + sqrt(dk[o] * dk[o] + rho));         // sigma < sqrt(d^2 + rho |z|^2), |z| = 1
        hi += 4.0 * KB_EPS * (dk[o] + hi);
        lo = 0.0;
        tau = hi * z2[o];                                       // exact if the other z vanish
        if (!(tau > lo && tau < hi)) tau = 0.5 * hi;
    }
    const double dorg = dk[o];
    bool conv = false;
    for (int it = 0; it < KB_DC_MAXIT; ++it) {
        double psi = 0.0, dpsi = 0.0, phi = 0.0, dphi = 0.0, asum = rho_inv;
        for (int j = 0; j <= i; ++j) {
            const double del = (dk[j] - dorg) - tau, w = (dk[j] + dorg) + tau;
            const double r = 1.0 / (del * w), tj = z2[j] * r;
            psi += tj; dpsi = fma(tj, r, dpsi); asum += fabs(tj);
        }
        for (int j = K - 1; j > i; --j) {
            const double del = (dk[j] - dorg) - tau, w = (dk[j] + dorg) + tau;
            const double r = 1.0 / (del * w), tj = z2[j] * r;
            phi += tj; dphi = fma(tj, r, dphi); asum += fabs(tj);
        }
        const double g = rho_inv + psi + phi;
        if (fabs(g) <= 4.0 * KB_EPS * asum) { conv = true; break; }
        if (g < 0.0) lo = tau; else hi = tau;
        const double sig = dorg + tau;
        const double a = ((dk[i] - dorg) - tau) * ((dk[i] + dorg) + tau);
        double eta = 0.0;
        bool have;
        if (!last) {
            const double b = ((dk[i + 1] - dorg) - tau) * ((dk[i + 1] + dorg) + tau);
            // "middle way": psi ~ s + S / (a - eta), phi ~ r + R / (b - eta), values and slopes matched at eta = 0
            const double S = dpsi * a * a, s0 = psi - dpsi * a, R = dphi * b * b, r0 = phi - dphi * b;
            have = dc_quad(rho_inv + s0 + r0, a, S, b, R, eta);
        } else {
            const double S = dpsi * a * a, c = rho_inv + psi - dpsi * a;
            have = c > 0.0;
            if (have) eta = a + S / c;
        }
        double nt = 0.5 * (lo + hi);
        if (have) {
            const double x2 = sig * sig + eta;
            if (x2 > 0.0) {
                const double cand = tau + eta / (sig + sqrt(x2));
                if (cand > lo && cand < hi) nt = cand;
            }
        }
        if (nt == tau || !(nt > lo && nt < hi)) { conv = true; break; }      // the bracket cannot shrink any further
        tau = nt;
    }
    o_out = o; tau_out = tau;
    return conv;
}

// ------------------------------------------------------------------------------------------------ merge: coefficients
KB_HD int dc_merge_scratch_bytes(int n) {
    return (int)sizeof(double) * 10 * (n + 2) + (int)sizeof(int) * 6 * (n + 2) + 256;
}

struct DcHdr {
    double c0, s0, org, tol, rho_inv;
    int K, nrot, fail;
};

// Builds CU (n x n) and CV ((n+sqre) x (n+sqre)) of node `nd` from the children's singular values and the boundary rows
// of their right vectors (arrays of depth + 1: src), writes the node's singular values (dst).  One workgroup.
// *info |= 1 if a secular root did not converge.
template <class C>
KB_HD void dc_merge_setup(const C& ctx, const double* d, const double* e, const DcWs& ws, DcNode nd, int src,
                          bool descending, int* info, double scale) {
    const int m = ws.m, n = nd.n, lo = nd.lo, sqre = nd.sqre, mc = n + sqre;
    const int nl = (n - 1) / 2, nr = n - nl - 1, ic = lo + nl;
    const double* Vs = ws.V[src];
    const double* Ds = ws.D[src];
    double* Dn = ws.D[src ^ 1];
    double* CU = ws.CU;
    double* CV = ws.CV;
    const int T = ctx.nthreads(), t = ctx.tid();
    // scratch carve
    double* dd = reinterpret_cast<double*>(ctx.scratch());
    double* z = dd + (n + 2);
    double* dk = z + (n + 2);
    double* zk = dk + (n + 2);          // z_k, then z_k^2 / rho
    double* tau = zk + (n + 2);
    double* vals = tau + (n + 2);
    double* rc = vals + (n + 2);
    double* rs = rc + (n + 2);
    double* nu = rs + (n + 2);
    double* nv = nu + (n + 2);
    int* ord = reinterpret_cast<int*>(nv + (n + 2));
    int* kd = ord + (n + 2);            // keep[0..K) from the front, deflated from the back
    int* rp = kd + (n + 2);
    int* rj = rp + (n + 2);
    int* oi = rj + (n + 2);
    int* pos = oi + (n + 2);
    DcHdr* H = reinterpret_cast<DcHdr*>(pos + (n + 2));
    const double alpha = d[ic] / scale, beta = e[ic] / scale;   // (e[m-1] is never the centre of a node)
    // ---- z and d in local order: 0 = centre / q column, 1..nl = child 1, nl+1.. = child 2
    for (int j = t; j < n; j += T) {
        double zz, dv;
        if (j == 0) { zz = 0.0; dv = 0.0; }
        else if (j <= nl) { zz = alpha * Vs[ic + (size_t)(lo + j - 1) * m]; dv = Ds[lo + j - 1]; }
        else { zz = beta * Vs[(ic + 1) + (size_t)(ic + 1 + (j - nl - 1)) * m]; dv = Ds[ic + 1 + (j - nl - 1)]; }
        z[j] = zz; dd[j] = dv;
    }
    double mx = 0.0;
    for (int j = t; j < n; j += T) mx = fmax(mx, fabs(j == 0 ? 0.0 : (j <= nl ? Ds[lo + j - 1] : Ds[ic + 1 + (j - nl - 1)])));
    mx = ctx.block_max(mx);
    if (t == 0) {
        const double a1 = alpha * Vs[ic + (size_t)ic * m];
        const double b2 = sqre ? beta * Vs[(ic + 1) + (size_t)(ic + 1 + nr) * m] : 0.0;
        double c0 = 1.0, s0 = 0.0, z0 = a1;
        if (sqre) {
            const double r0 = hypot(a1, b2);
            if (r0 > 0.0) { c0 = a1 / r0; s0 = b2 / r0; }
            z0 = r0;
        }
        z[0] = z0;
        double org = fmax(fmax(fabs(alpha), fabs(beta)), mx);
        if (!(org > 0.0)) org = 1.0;
        H->c0 = c0; H->s0 = s0; H->org = org;
        H->tol = 8.0 * KB_EPS * fmax(fmax(fabs(alpha), fabs(beta)) / org, mx / org);
        H->fail = 0;
    }
    ctx.sync();
    {
        const double inv = 1.0 / H->org;
        for (int j = t; j < n; j += T) { dd[j] *= inv; z[j] *= inv; }
    }
    ctx.sync();
    // ---- sorted order of the locals 1..n-1 by d (ties by index); ord[0] = 0
    for (int j = t; j < n; j += T) {
        if (j == 0) { ord[0] = 0; continue; }
        int r = 1;
        const double dj = dd[j];
        for (int k = 1; k < n; ++k) r += (dd[k] < dj || (dd[k] == dj && k < j)) ? 1 : 0;
        ord[r] = j;
    }
    ctx.sync();
    // ---- deflation scan (serial)
    if (t == 0) {
        const double tol = H->tol;
        if (fabs(z[0]) <= tol) z[0] = tol;
        int K = 0, nd_ = 0, nrot = 0, prev = -1;
        kd[K++] = 0;
        for (int s = 1; s < n; ++s) {
            const int j = ord[s];
            if (fabs(z[j]) <= tol) { kd[n - 1 - nd_++] = j; continue; }
            if (prev >= 0 && dd[j] - dd[prev] <= tol) {
                // rotate (prev, j): z_prev -> 0; prev leaves the secular problem with its d
                const double sv = z[prev], cv = z[j];
                const double tt = hypot(cv, sv);
                rc[nrot] = cv / tt; rs[nrot] = -sv / tt; rp[nrot] = prev; rj[nrot] = j; ++nrot;
                z[j] = tt; z[prev] = 0.0;
                --K;                                    // prev was the last kept entry
                kd[n - 1 - nd_++] = prev;
            }
            kd[K++] = j;
            prev = j;
        }
        double rho = 0.0;
        for (int k = 0; k < K; ++k) { dk[k] = dd[kd[k]]; zk[k] = z[kd[k]]; rho += zk[k] * zk[k]; }
        if (K > 1 && dk[1] <= 0.5 * tol) dk[1] = 0.5 * tol;
        H->K = K; H->nrot = nrot; H->rho_inv = 1.0 / rho;
    }
    ctx.sync();
    const int K = H->K, nrot = H->nrot;
    const double rho_inv = H->rho_inv;
    for (int k = t; k < K; k += T) { vals[k] = zk[k]; }              // vals: a copy of the signed z_k (sign source)
    ctx.sync();
    const double vals_sign0 = vals[0];
    for (int k = t; k < K; k += T) zk[k] = zk[k] * zk[k] * rho_inv;    // normalised squares
    ctx.sync();
    // ---- secular roots, one per thread
    int bad = 0;
    for (int i = t; i < K; i += T) {
        int o; double tu;
        if (!dc_secular(i, K, dk, zk, rho_inv, o, tu)) bad = 1;
        oi[i] = o; tau[i] = tu;
    }
    bad = ctx.block_max(bad);
    if (t == 0 && bad) *info |= 1;
    // ---- Loewner: z_j^2 = prod_i (sigma_i^2 - d_j^2) / prod_{i != j} (d_i^2 - d_j^2), sign of the original z_j
    for (int j = t; j < K; j += T) {
        const double dj = dk[j];
        double v;
        { const int o = oi[K - 1]; v = ((dj - dk[o]) - tau[K - 1]) * ((dj + dk[o]) + tau[K - 1]); }
        for (int i = 0; i < j; ++i) {
            const int o = oi[i];
            v *= ((dj - dk[o]) - tau[i]) * ((dj + dk[o]) + tau[i]) / (dj - dk[i]) / (dj + dk[i]);
        }
        for (int i = j; i < K - 1; ++i) {
            const int o = oi[i];
            v *= ((dj - dk[o]) - tau[i]) * ((dj + dk[o]) + tau[i]) / (dj - dk[i + 1]) / (dj + dk[i + 1]);
        }
        const double zh = sqrt(fabs(v));
        z[j] = (vals[j] < 0.0) ? -zh : zh;                               // z: now zhat (K entries)
    }
    ctx.sync();
    // ---- column norms of the K x K vector matrices
    for (int i = t; i < K; i += T) {
        const int o = oi[i];
        const double dorg = dk[o], tu = tau[i];
        double su = 1.0, sv = 0.0;                                      // u_0 = -1
        for (int k = 0; k < K; ++k) {
            const double v = z[k] / (((dk[k] - dorg) - tu) * ((dk[k] + dorg) + tu));
            sv = fma(v, v, sv);
            if (k > 0) { const double u = dk[k] * v; su = fma(u, u, su); }
        }
        nu[i] = 1.0 / sqrt(su); nv[i] = 1.0 / sqrt(sv);
    }
    ctx.sync();
    // ---- all n singular values (scaled) and their output positions
    for (int k = t; k < n; k += T) vals[k] = (k < K) ? dk[oi[k]] + tau[k] : dd[kd[k]];
    ctx.sync();
    for (int k = t; k < n; k += T) {
        int r = 0;
        const double vk = vals[k];
        if (descending) { for (int q = 0; q < n; ++q) r += (vals[q] > vk || (vals[q] == vk && q < k)) ? 1 : 0; }
        else { for (int q = 0; q < n; ++q) r += (vals[q] < vk || (vals[q] == vk && q < k)) ? 1 : 0; }
        pos[k] = r;
        Dn[lo + r] = vk * H->org;
    }
    ctx.sync();
    // ---- coefficient matrices.  Raw basis row of local j >= 1: j - 1 (child 1) or j (child 2); local 0: U row nl,
    // V rows nl (c0) and n (s0, sqre only).  Zero, fill, then the deflation rotations on rows (reverse order).
    double* CUb = CU + lo + (size_t)lo * m;
    double* CVb = CV + lo + (size_t)lo * m;
    for (int x = t; x < mc * mc; x += T) {
        const int r = x % mc, c = x / mc;
        CVb[r + (size_t)c * m] = 0.0;
        if (r < n && c < n) CUb[r + (size_t)c * m] = 0.0;
    }
    ctx.sync();
    const double c0 = H->c0, s0 = H->s0;
    for (int x = t; x < K * K; x += T) {
        const int k = x % K, i = x / K;
        const int o = oi[i];
        const double v = z[k] / (((dk[k] - dk[o]) - tau[i]) * ((dk[k] + dk[o]) + tau[i]));
        double vv = v * nv[i], uu = (k == 0 ? -1.0 : dk[k] * v) * nu[i];
        if (K == 1) { vv = (vals_sign0 < 0.0) ? 1.0 : -1.0; uu = -1.0; }    // the 1 x 1 problem [z_0] (z_0 may be 0)
        const int col = pos[i];
        const int j = kd[k];
        if (j == 0) {
            CUb[nl + (size_t)col * m] = uu;
            CVb[nl + (size_t)col * m] = c0 * vv;
            if (sqre) CVb[n + (size_t)col * m] = s0 * vv;
        } else {
            const int row = (j <= nl) ? j - 1 : j;
            CUb[row + (size_t)col * m] = uu;
            CVb[row + (size_t)col * m] = vv;
        }
    }
    for (int k = K + t; k < n; k += T) {                   // deflated: unit columns
        const int j = kd[k], row = (j <= nl) ? j - 1 : j, col = pos[k];
        CUb[row + (size_t)col * m] = 1.0;
        CVb[row + (size_t)col * m] = 1.0;
    }
    if (sqre && t == 0) { CVb[nl + (size_t)n * m] = -s0; CVb[n + (size_t)n * m] = c0; }
    ctx.sync();
    if (nrot > 0) {
        for (int c = t; c < n; c += T) {
            double* cu = CUb + (size_t)c * m;
            double* cv = CVb + (size_t)c * m;
            for (int q = nrot - 1; q >= 0; --q) {
                const int jp = rp[q], jj = rj[q];
                const int a = (jp <= nl) ? jp - 1 : jp, b = (jj <= nl) ? jj - 1 : jj;
                const double cq = rc[q], sq = rs[q];
                // basis columns were rotated  x' = c x + s y, y' = c y - s x  (x = prev, y = j): on the coefficients
                // row_prev' = c row_prev - s row_j,  row_j' = s row_prev + c row_j
                double x = cu[a], y = cu[b];
                cu[a] = cq * x - sq * y; cu[b] = sq * x + cq * y;
                x = cv[a]; y = cv[b];
                cv[a] = cq * x - sq * y; cv[b] = sq * x + cq * y;
            }
        }
    }
    ctx.sync();
}

// Basis accessors of a merge (block diagonal in the arrays of depth + 1); r, k local to the node.
KB_HD double dc_ubasis(const double* Us, int m, int lo, int nl, int r, int k) {
    if (r < nl) return (k < nl) ? Us[(lo + r) + (size_t)(lo + k) * m] : 0.0;
    if (r == nl) return (k == nl) ? 1.0 : 0.0;
    return (k > nl) ? Us[(lo + r) + (size_t)(lo + k) * m] : 0.0;
}
KB_HD double dc_vbasis(const double* Vs, int m, int lo, int nl, int r, int k) {
    if (r <= nl) return (k <= nl) ? Vs[(lo + r) + (size_t)(lo + k) * m] : 0.0;
    return (k > nl) ? Vs[(lo + r) + (size_t)(lo + k) * m] : 0.0;
}

// Reference (host) form of the node products  U = Ubasis CU,  V = Vbasis CV  (the device runs them as MFMA tiles).
inline void dc_merge_apply_ref(const DcWs& ws, DcNode nd, int src) {
    const int m = ws.m, n = nd.n, lo = nd.lo, mc = n + nd.sqre, nl = (n - 1) / 2;
    const int dst = src ^ 1;
    for (int c = 0; c < mc; ++c)
        for (int r = 0; r < mc; ++r) {
            double sv = 0.0, su = 0.0;
            for (int k = 0; k < mc; ++k) sv += dc_vbasis(ws.V[src], m, lo, nl, r, k) * ws.CV[(lo + k) + (size_t)(lo + c) * m];
            ws.V[dst][(lo + r) + (size_t)(lo + c) * m] = sv;
            if (r < n && c < n) {
                for (int k = 0; k < n; ++k) su += dc_ubasis(ws.U[src], m, lo, nl, r, k) * ws.CU[(lo + k) + (size_t)(lo + c) * m];
                ws.U[dst][(lo + r) + (size_t)(lo + c) * m] = su;
            }
        }
}

}  // namespace kb
