// All eigenvalues of the reduced matrix's Hessenberg form by simultaneous Ehrlich-Aberth iteration on Hyman's
// recurrence, organised as divide and conquer - the fast path of the zgeev replacement (reference kbdm.py:192); the
// multishift QR iteration (kb_hqr2.hpp) stays as the fallback for members this path declines.
//
//   * Hyman: for a shift z solve the rows n-1 .. 1 of (H - z I) x = 0 upwards from x_{n-1} = 1; row 0 then gives
//     rho(z) = sum_j (H - z I)_{0j} x_j, proportional to det(H - z I); the same recurrence differentiated gives rho'(z).
//     Backward stable for Hessenberg matrices (Wilkinson).  For ALL roots of a node at once the recurrence is a
//     triangular-solve-shaped product H X with one column per root: blocks of 32 rows, the part of a block's rows
//     that multiplies already finished rows of X is a dense product (FP64 MFMA on the device), the 32 x 32 triangle
//     inside the block is a short serial recurrence.  Columns are rescaled by powers of two at block ends (only the
//     ratio rho / rho' is used).
//   * Ehrlich-Aberth: z_i <- z_i - N_i / (1 - N_i sum_{j != i} 1 / (z_i - z_j)),  N_i = rho / rho' (z_i): cubically
//     convergent, the repulsion term keeps two approximations from settling on one eigenvalue.
//   * Divide and conquer: the two diagonal halves of a Hessenberg matrix are Hessenberg; their eigenvalues are the
//     starting values of the parent (separated deterministically).  On reduced KBDM matrices that needs 5-8 iterations
//     per root and level (tools/proto_aberth_dc.py: noise-free and noisy signals, l < m, q > 0, p = 2), ~8 n^3 complex
//     multiply-adds in all - the flops of the QR iteration, but as independent columns of a matrix product instead of
//     a chain of bulge chases on one CU.  Leaves (<= 32 rows): the one-wavefront solver of kb_hqr2.hpp.
//   * A member leaves this path for the QR iteration when a subdiagonal is negligible (the matrix splits: Hyman's
//     division), a root does not settle within the iteration budget, or the first two power sums of the roots miss
//     trace(H) / trace(H^2).
#pragma once
#include "kb_complex.hpp"
#if !defined(__HIP_DEVICE_COMPILE__)
#include <vector>
#endif

namespace kb {

constexpr int KB_AB_LEAF = 32;        // largest leaf
constexpr int KB_AB_BUDGET = 24;      // iterations per level (even: the root buffers alternate)
constexpr int KB_AB_TILE = 64;        // roots per workgroup
constexpr int KB_AB_BLK = 32;         // rows per block of the recurrence
constexpr int KB_AB_INNER_BUDGET = 4;  // iterations of a level below the root of the tree (even: the root buffers alternate)
static_assert(KB_AB_INNER_BUDGET % 2 == 0, "the root buffers alternate per launch: a level must end on the buffer it started on");
// The tail of the root level: once a member has at most KB_AB_TAIL_ROOTS unsettled roots (and the tile iterations have had
// KB_AB_TAIL_FROM launches), every remaining root gets a wavefront of its own (k_ab_tail: Hyman's recurrence column by
// column, the running sums in registers) instead of a 64-root MFMA tile that costs the same whether it carries one root
// or 64.  Members of more than KB_AB_TAIL_MAXL rows stay on the tile kernel (the register-resident sums).
constexpr int KB_AB_TAIL_FROM = 8;
constexpr int KB_AB_TAIL_ROOTS = 48;
constexpr int KB_AB_TAIL_WGS = 12;     // workgroups of four wavefronts per member: one root per wavefront
constexpr int KB_AB_TAIL_MAXC = 8;
constexpr int KB_AB_TAIL_MAXL = 64 * KB_AB_TAIL_MAXC;
KB_HD bool ab_tail_takes(int l, int depth, int iter, int nactive) {
    return depth == 0 && iter >= KB_AB_TAIL_FROM && l <= KB_AB_TAIL_MAXL && nactive <= KB_AB_TAIL_ROOTS;
}

struct AbNode { int a, n; };

KB_HD int ab_depth(int l) {
    int D = 0, s = l;
    while (s > KB_AB_LEAF) { s = s - s / 2; ++D; }      // the larger half
    return D;
}
KB_HD AbNode ab_node(int l, int depth, int idx) {
    AbNode nd; nd.a = 0; nd.n = l;
    for (int b = depth - 1; b >= 0; --b) {
        const int h = nd.n / 2;
        if ((idx >> b) & 1) { nd.a += h; nd.n -= h; }
        else nd.n = h;
    }
    return nd;
}
KB_HD int ab_level_nmax(int l, int depth) {              // largest node of a level
    int n = l;
    for (int k = 0; k < depth; ++k) n = n - n / 2;
    return n;
}

// Per-member workspace (doubles): roots z[2][l] (complex) and settled flags conv[2][l] (ints) of alternate iterations,
// last correction |dz| (l), the panels of the recurrence: [rows][128] complex, l * ceil(l / 64) rows.
struct AbWs {
    int l;
    cd* z[2];
    double* lastc;
    int* conv[2];
    int* nact;          // unsettled roots of the root node as k_ab_iter counted them in its last launch (read by k_ab_tail)
    cd* panel;
};
KB_HD long long ab_ws_doubles(int l) {
    return 4LL * l + l + l + 16 + 256LL * l * ((l + KB_AB_TILE - 1) / KB_AB_TILE);
}
KB_HD AbWs ab_ws(double* base, int l) {
    AbWs w; w.l = l;
    w.z[0] = reinterpret_cast<cd*>(base);
    w.z[1] = w.z[0] + l;
    w.lastc = base + 4 * (size_t)l;
    w.conv[0] = reinterpret_cast<int*>(w.lastc + l);
    w.conv[1] = w.conv[0] + l;
    w.nact = reinterpret_cast<int*>(base + 6 * (size_t)l);        // (the 16 spare doubles in front of the panels)
    w.panel = reinterpret_cast<cd*>(base + ((6 * (size_t)l + 16 + 1) & ~(size_t)1));
    return w;
}

// deterministic separation of the starting values of a node (coincident eigenvalues of the two halves)
KB_HD cd ab_perturb(cd z, int j, double hnorm) {
    const double twopi = 6.283185307179586476925286766559;
    const double a1 = twopi * (j * 0.61803398875), a2 = twopi * (j * 0.754877666);
    const cd f = mk(1.0 + 1e-9 * cos(a1), 1e-9 * sin(a1));
    return z * f + mk(1e-12 * hnorm * cos(a2), 1e-12 * hnorm * sin(a2));
}

KB_HD cd ab_recip(cd a) {
    return cdiv(mk(1.0, 0.0), a);
}

KB_HD bool ab_finite(cd a) { return (a.x - a.x == 0.0) && (a.y - a.y == 0.0); }

// One Aberth update of root i of a node from its Newton correction N = rho / rho' and S = sum_{j != i} 1 / (z_i - z_j).
// Returns the new root; *dz = |correction| (infinity if the step is not finite: the root stays).
KB_HD cd ab_update(cd z, cd rho, cd rhop, cd S, double* dz) {
    const cd N = cdiv(rho, rhop);
    const cd den = mk(1.0, 0.0) - N * S;
    const cd corr = cdiv(N, den);
    if (!ab_finite(corr)) { *dz = 1.79769313486231570815e308; return z; }
    *dz = cabs(corr);
    return z - corr;
}
// A root has settled when its correction falls below 1e-10 |z|: the iteration converges cubically, so the root it
// leaves behind is accurate to (correction)^3 / (separation)^2 - far below working precision for any separation that
// working precision resolves; waiting for a correction of a few ulp costs every root one more iteration and roots
// whose evaluation noise exceeds an ulp the whole budget.  A root that has not settled when the budget ends is still
// accepted if its last correction is below 1e-9 |z| (noise-limited, condition number ~1e6 and beyond).
KB_HD bool ab_converged(double dz, cd z, double hnorm) { return dz <= 1e-10 * fmax(cabs(z), 1e-6 * hnorm); }
// below the root of the tree a node's eigenvalues are only the parent's starting values, which the coupling element moves
// by 1e-2 anyway: a correction of 1e-3 ends the iteration of a root there (measured on C2: tolerances from 1e-6 to 8e-3 leave
// the iterations of the root level unchanged, 3e-2 costs it half as many again)
KB_HD bool ab_converged_inner(double dz, cd z, double hnorm) { return dz <= 1e-3 * fmax(cabs(z), 1e-6 * hnorm); }
KB_HD bool ab_acceptable(double dz, cd z, double hnorm) { return dz <= 1e-9 * fmax(cabs(z), 1e-6 * hnorm); }

// The last correction dz certifies a root to working precision when dz^3 / sep^2 (the error a cubically convergent step
// leaves behind, sep = distance to the nearest other root) is below 1e-17 of the root's scale.  sep2 = sep^2.
KB_HD bool ab_certified(double dz, cd z, double hnorm, double sep2) {
    const double sc = fmax(cabs(z), 1e-6 * hnorm);
    const double r = dz / sc;
    return r * r * r <= 1e-17 * (sep2 / (sc * sc));
}

// A subdiagonal entry that small splits the matrix: Hyman's recurrence divides by it (the QR iteration deflates there)
KB_HD bool ab_negligible_sub(cd hsub, cd hk, cd hk1) {
    return cabs1(hsub) <= 64.0 * KB_ULP * (cabs1(hk) + cabs1(hk1)) || cabs1(hsub) <= 1e-280;
}

#if !defined(__HIP_DEVICE_COMPILE__)
// ---- host reference of the whole solver (tests/hostsim): the same tree, starting values, update and acceptance rules
// as the device kernels; the recurrence is evaluated row by row (the device's blocked MFMA form sums in another order).
// H: n x n column-major (ld); w: n eigenvalues out; returns 0, or 1 if the member would go to the QR iteration.
inline void ab_host_newton(const cd* H, int ld, int a, int n, const cd* z, int R, cd* rho, cd* rhop, cd* X, cd* Y) {
    // X, Y: n x R work (row-major by row of the recurrence)
    for (int c = 0; c < R; ++c) { X[(size_t)(n - 1) * R + c] = mk(1.0, 0.0); Y[(size_t)(n - 1) * R + c] = czero(); }
    for (int k = n - 1; k >= 0; --k) {
        cd inv = czero();
        if (k > 0) inv = ab_recip(H[(a + k) + (size_t)(a + k - 1) * ld]);
        for (int c = 0; c < R; ++c) {
            cd s = czero(), sp = czero();
            for (int j = k; j < n; ++j) {
                const cd h = H[(a + k) + (size_t)(a + j) * ld];
                cfma(s, h, X[(size_t)j * R + c]);
                cfma(sp, h, Y[(size_t)j * R + c]);
            }
            s = s - z[c] * X[(size_t)k * R + c];
            sp = sp - z[c] * Y[(size_t)k * R + c] - X[(size_t)k * R + c];
            if (k > 0) { X[(size_t)(k - 1) * R + c] = -(s * inv); Y[(size_t)(k - 1) * R + c] = -(sp * inv); }
            else { rho[c] = s; rhop[c] = sp; }
        }
        if (k > 0 && ((n - k) % KB_AB_BLK) == 0)        // rescale the columns by powers of two
            for (int c = 0; c < R; ++c) {
                const double mx = fmax(fabs(X[(size_t)(k - 1) * R + c].x), fabs(X[(size_t)(k - 1) * R + c].y));
                int e = 0;
                if (mx > 0.0 && mx == mx) frexp(mx, &e);
                if (e > 60 || e < -60) {
                    const double f = ldexp(1.0, -e);
                    for (int j = k - 1; j < n; ++j) { X[(size_t)j * R + c] = f * X[(size_t)j * R + c]; Y[(size_t)j * R + c] = f * Y[(size_t)j * R + c]; }
                }
            }
    }
}

template <class LeafSolver>
inline int ab_host_eig(const cd* H, int ld, int n, double hnorm, cd* w, LeafSolver leaf, long long* iters_out = nullptr) {
    for (int k = 1; k < n; ++k)
        if (ab_negligible_sub(H[k + (size_t)(k - 1) * ld], H[k + (size_t)k * ld], H[(k - 1) + (size_t)(k - 1) * ld])) return 1;
    const int D = ab_depth(n);
    std::vector<cd> z(n), zn(n), rho(n), rhop(n), X, Y;
    std::vector<double> lastc(n, 0.0);
    std::vector<int> conv(n, 0);
    long long iters = 0;
    for (int idx = 0; idx < (1 << D); ++idx) {
        const AbNode nd = ab_node(n, D, idx);
        if (leaf(H, ld, nd.a, nd.n, z.data() + nd.a)) return 1;
    }
    for (int depth = D - 1; depth >= 0; --depth)
        for (int idx = 0; idx < (1 << depth); ++idx) {
            const AbNode nd = ab_node(n, depth, idx);
            cd* zz = z.data() + nd.a;
            for (int j = 0; j < nd.n; ++j) { zz[j] = ab_perturb(zz[j], nd.a + j, hnorm); conv[nd.a + j] = 0; }
            X.assign((size_t)nd.n * nd.n, czero());
            Y.assign((size_t)nd.n * nd.n, czero());
            for (int it = 0; it < KB_AB_BUDGET; ++it) {
                bool any = false;
                for (int j = 0; j < nd.n; ++j) any = any || !conv[nd.a + j];
                if (!any) break;
                ab_host_newton(H, ld, nd.a, nd.n, zz, nd.n, rho.data(), rhop.data(), X.data(), Y.data());
                for (int i = 0; i < nd.n; ++i) {
                    zn[i] = zz[i];
                    if (conv[nd.a + i]) continue;
                    cd S = czero();
                    for (int j = 0; j < nd.n; ++j)
                        if (j != i) S = S + ab_recip(zz[i] - zz[j]);
                    double dz;
                    zn[i] = ab_update(zz[i], rho[i], rhop[i], S, &dz);
                    lastc[nd.a + i] = dz;
                    if (ab_converged(dz, zn[i], hnorm)) conv[nd.a + i] = 1;
                    ++iters;
                }
                for (int i = 0; i < nd.n; ++i) zz[i] = zn[i];
            }
            for (int j = 0; j < nd.n; ++j)
                if (!ab_acceptable(lastc[nd.a + j], zz[j], hnorm)) return 1;
        }
    for (int k = 0; k < n && n > 1; ++k) {                  // (the root level's roots: certified by their separation)
        double sep2 = 1.79769313486231570815e308;
        for (int j = 0; j < n; ++j)
            if (j != k) { const cd dd = z[k] - z[j]; sep2 = fmin(sep2, dd.x * dd.x + dd.y * dd.y); }
        if (!ab_certified(lastc[k], z[k], hnorm, sep2)) return 1;
    }
    if (iters_out) *iters_out = iters;
    // power sums against the traces
    cd t1 = czero(), t2 = czero(), s1 = czero(), s2 = czero();
    double a1 = 0.0, a2 = 0.0;
    for (int k = 0; k < n; ++k) {
        const cd d = H[k + (size_t)k * ld];
        t1 = t1 + d; t2 = t2 + d * d;
        if (k + 1 < n) t2 = t2 + 2.0 * (H[(k + 1) + (size_t)k * ld] * H[k + (size_t)(k + 1) * ld]);
        s1 = s1 + z[k]; s2 = s2 + z[k] * z[k];
        a1 += cabs(z[k]); a2 += abs2(z[k]);
    }
    if (cabs(s1 - t1) > 1e-9 * (a1 + 1e-300) || cabs(s2 - t2) > 1e-9 * (a2 + 1e-300)) return 1;
    for (int k = 0; k < n; ++k) w[k] = z[k];
    return 0;
}
#endif

}  // namespace kb
