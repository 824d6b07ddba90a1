// CPU debugging harness for the per-member algorithms in llckbdm_amd/csrc/kb_*.hpp.
//
// TEST INFRASTRUCTURE ONLY.  It instantiates the very same templates the HIP kernels
// instantiate, but with kb::HostCtx (a one-thread "workgroup"), so that indexing and
// convergence logic can be checked against the oracle on the build box, which has no GPU.
// It is compiled by tests/test_hostsim.py into tests/hostsim/_build/ and is never loaded
// by the llckbdm_amd package (which talks to the HIP library only and fails loudly
// without it).
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kb_eig.hpp"
#include "kb_hqr_ms.hpp"
#include "kb_hqr2.hpp"
#include "kb_svd.hpp"
#include "kb_bdsdc.hpp"
#include "kb_aberth.hpp"
#include "kb_panel_team.hpp"

#include <functional>
#include <pthread.h>
#include <thread>

using namespace kb;
typedef std::complex<double> zc;
constexpr int HS_MAXC = 2048;   // host wave size is 1: one register chunk per row

static HostCtx make_ctx(std::vector<char>& arena, size_t bytes) {
    arena.assign(bytes + KB_RED_BYTES + 64, 0);
    HostCtx c;
    c.smem = arena.data();
    c.smem_bytes = (int)arena.size();
    return c;
}

// A team of T one-thread "workgroups" (kb_team.hpp): the T roles run as threads that meet in a pthread barrier; every
// role has a scratch arena of its own, the matrices and the exchange buffer are shared - what the device does.
static int hs_team_size() {
    const char* tz = getenv("HS_PANEL_T");
    const int T = tz ? atoi(tz) : 1;
    return T < 1 ? 1 : T;
}
static void hs_run_team(int T, size_t scratch_bytes, cd* xbuf, const std::function<void(HostCtx&, PanelTeam<HostCtx>&)>& body) {
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, nullptr, T);
    auto role_main = [&](int role) {
        std::vector<char> arena;
        HostCtx c = make_ctx(arena, scratch_bytes);
        PanelTeam<HostCtx> tm;
        tm.T = T; tm.role = role; tm.epoch = 0; tm.ctl = nullptr; tm.xb = xbuf; tm.failed = 0;
        tm.barrier = [](void* b) { pthread_barrier_wait(static_cast<pthread_barrier_t*>(b)); };
        tm.barg = &bar;
        body(c, tm);
    };
    std::vector<std::thread> th;
    for (int r = 1; r < T; ++r) th.emplace_back(role_main, r);
    role_main(0);
    for (auto& t : th) t.join();
    pthread_barrier_destroy(&bar);
}
constexpr int HS_ZR = 8;       // rows of the row-product partial sums per batch (host wave size 1: 8 chunks per batch)

// blocked bidiagonalisation exactly as the kernels sequence it: panels + trailing updates + tail
static void hs_bidiag_blocked(HostCtx& ctx, int m, cd* A, double* d, double* e, cd* tq, cd* tp, cd* UR) {
    const int npan = bidiag_num_panels(m);
    std::vector<cd> X((size_t)m * KB_NB), Y((size_t)m * KB_NB), xbuf((size_t)panel_team_xbuf_elems(m));
    for (int p = 0; p < npan; ++p) {
        const int p0 = p * KB_NB, n = m - p0;
        cd* Ab = A + p0 + (size_t)p0 * m;
        cd* Ub = UR + p0 + (size_t)p0 * m;
        // the team panel the device runs (HS_PANEL_T "workgroups" per member)
            hs_run_team(hs_team_size(), panel_team_scratch_bytes(n, HS_ZR, 1), xbuf.data(), [&](HostCtx& c, PanelTeam<HostCtx>& tm) {
                bidiag_panel_team(c, tm, n, Ab, m, d + p0, e + p0, tq + p0, tp + p0, Ub, m, X.data(), Y.data(), m, HS_ZR);
            });
        for (int c = KB_NB; c < n; ++c)
            for (int r = KB_NB; r < n; ++r) {
                cd acc = Ab[r + (size_t)c * m];
                for (int t = 0; t < KB_NB; ++t)
                    acc = acc - Ab[r + (size_t)t * m] * conj(Y[c + (size_t)t * m]) - X[r + (size_t)t * m] * conj(Ub[c + (size_t)t * m]);
                Ab[r + (size_t)c * m] = acc;
            }
    }
    bidiag(ctx, m, A, m, d, e, tq, tp, UR, m, npan * KB_NB);
}

// Eigenvalues of a Hessenberg matrix by the Ehrlich-Aberth path (kb_aberth.hpp: host reference), leaves by the small
// QR iteration; returns 0, or 1 if the member would fall back to the QR iteration.
static int hs_aberth_path(HostCtx& ctx, int n, const cd* Hc, double hnorm, cd* mu, long long* iters) {
    auto leaf = [&](const cd* H, int ld, int a, int nn, cd* z) -> int {
        std::vector<cd> S((size_t)nn * nn);
        for (int c = 0; c < nn; ++c)
            for (int r = 0; r < nn; ++r) S[r + (size_t)c * nn] = (r <= c + 1) ? H[(a + r) + (size_t)(a + c) * ld] : czero();
        int info = 0;
        WaveCtx<HostCtx> wc{ctx, nullptr, 0};
        hqr_eigvals(wc, nn, S.data(), nn, z, &info);
        return info;
    };
    return ab_host_eig(Hc, n, n, hnorm, mu, leaf, iters);
}

extern "C" {

// Bidiagonal divide and conquer (kb_bdsdc.hpp) exactly as the device sequences it: leaves, then one depth after the
// other (setup + the two products per node).  d, e: m and m-1 (e is read up to m-1 entries); X, Y: m x m column-major;
// s descending.  Returns the info word (1 = a secular root hit the iteration limit).
int hs_bdsdc(const double* d_in, const double* e_in, int m, double* X, double* s, double* Y) {
    std::vector<double> d(d_in, d_in + m), e(m, 0.0);
    for (int i = 0; i + 1 < m; ++i) e[i] = e_in[i];
    std::vector<double> wsb(dc_ws_doubles(m), 0.0);
    DcWs ws = dc_ws(wsb.data(), m);
    std::vector<char> arena;
    HostCtx ctx = make_ctx(arena, dc_merge_scratch_bytes(m) + dc_leaf_scratch_bytes(KB_DC_LEAF));
    const int L = dc_depth(m);
    int info = 0;
    const double scale = dc_scale(ctx, d.data(), e.data(), m);
    for (int idx = 0; idx < (1 << L); ++idx) {
        const DcNode nd = dc_node(m, L, idx);
        dc_leaf(ctx, d.data(), e.data(), m, nd, ws.U[L & 1], ws.V[L & 1], ws.D[L & 1], L == 0, scale);
    }
    for (int depth = L - 1; depth >= 0; --depth) {
        const int src = (depth + 1) & 1;
        for (int idx = 0; idx < (1 << depth); ++idx) {
            const DcNode nd = dc_node(m, depth, idx);
            dc_merge_setup(ctx, d.data(), e.data(), ws, nd, src, depth == 0, &info, scale);
            dc_merge_apply_ref(ws, nd, src);
        }
    }
    memcpy(X, ws.U[0], sizeof(double) * m * m);
    memcpy(Y, ws.V[0], sizeof(double) * m * m);
    for (int i = 0; i < m; ++i) s[i] = ws.D[0][i] * scale;
    return info;
}

// A (m x m column-major) -> L (m x m), s (m), R (m x m), A = L diag(s) R^H
int hs_svd(const double* A_in, int m, double* L_out, double* s_out, double* R_out) {
    std::vector<cd> A(m * m), Q(m * m), P(m * m), UR(m * m), tq(m), tp(m);
    std::vector<double> d(m), e(m);
    memcpy(A.data(), A_in, sizeof(cd) * m * m);
    std::vector<char> arena;
    HostCtx ctx = make_ctx(arena, panel_team_scratch_bytes(m, HS_ZR, 1) + 4 * m);
    hs_bidiag_blocked(ctx, m, A.data(), d.data(), e.data(), tq.data(), tp.data(), UR.data());
    gen_unitary_cols<HostCtx, HS_MAXC>(ctx, m, m, 0, A.data(), m, tq.data(), Q.data(), m, 0, m);
    gen_unitary_cols<HostCtx, HS_MAXC>(ctx, m, m - 1, 1, UR.data(), m, tp.data(), P.data(), m, 0, m);
    // divide and conquer on (d, e), then L = Q X, R = P Y (what the device does: kb_bdsdc.hpp + real GEMMs)
    std::vector<double> X((size_t)m * m), Y((size_t)m * m);
    const int info = hs_bdsdc(d.data(), e.data(), m, X.data(), s_out, Y.data());
    cd* Lo = reinterpret_cast<cd*>(L_out);
    cd* Ro = reinterpret_cast<cd*>(R_out);
    for (int c = 0; c < m; ++c)
        for (int r = 0; r < m; ++r) {
            cd l = czero(), rr = czero();
            for (int k = 0; k < m; ++k) {
                l = l + X[k + (size_t)c * m] * Q[r + (size_t)k * m];
                rr = rr + Y[k + (size_t)c * m] * P[r + (size_t)k * m];
            }
            Lo[r + (size_t)c * m] = l; Ro[r + (size_t)c * m] = rr;
        }
    return info;
}

// bidiagonalisation only: returns d, e and explicit Q, P (for stage debugging)
int hs_bidiag(const double* A_in, int m, double* d, double* e, double* Q_out, double* P_out) {
    std::vector<cd> A(m * m), UR(m * m), tq(m), tp(m);
    memcpy(A.data(), A_in, sizeof(cd) * m * m);
    std::vector<char> arena;
    HostCtx ctx = make_ctx(arena, panel_team_scratch_bytes(m, HS_ZR, 1));
    hs_bidiag_blocked(ctx, m, A.data(), d, e, tq.data(), tp.data(), UR.data());
    gen_unitary_cols<HostCtx, HS_MAXC>(ctx, m, m, 0, A.data(), m, tq.data(), reinterpret_cast<cd*>(Q_out), m, 0, m);
    gen_unitary_cols<HostCtx, HS_MAXC>(ctx, m, m - 1, 1, UR.data(), m, tp.data(), reinterpret_cast<cd*>(P_out), m, 0, m);
    return 0;
}

// W (n x n column-major) -> mu (n), P (n x n, column k = eigenvector of mu[k])
int hs_eig(const double* W_in, int n, double* mu_out, double* P_out) {
    std::vector<cd> W(n * n), Qh(n * n), Hc(n * n), Ht(n * n), X(n * n), th(n);
    memcpy(W.data(), W_in, sizeof(cd) * n * n);
    std::vector<char> arena;
    HostCtx ctx = make_ctx(arena, gehd2_scratch_bytes(n, 1, 1) + panel_team_scratch_bytes(n, HS_ZR, 1) + invit_scratch_bytes_per_wave(n) +
                                      hqr2_scratch_bytes(KB2_WIN_DEV));
    {   // blocked Hessenberg reduction as the kernels sequence it: panels + updates + tail
        const int npan = bidiag_num_panels(n);
        std::vector<cd> Yp((size_t)n * KB_NB), Zp((size_t)n * KB_NB), VTp((size_t)n * KB_NB), MTp(KB_NB * KB_NB);
        for (int pnl = 0; pnl < npan; ++pnl) {
            const int p0 = pnl * KB_NB;
            {
                std::vector<cd> xbuf((size_t)panel_team_xbuf_elems(n));
                hs_run_team(hs_team_size(), panel_team_scratch_bytes(n, HS_ZR, 1), xbuf.data(), [&](HostCtx& c, PanelTeam<HostCtx>& tm) {
                    hess_panel_team(c, tm, n, W.data(), n, p0, th.data() + p0, Yp.data(), n, VTp.data(), n, MTp.data(), HS_ZR);
                });
            }
            hess_ytop_block(ctx, n, W.data(), n, p0, VTp.data(), n, Yp.data(), n);
            hess_z_block(ctx, n, W.data(), n, p0, VTp.data(), n, MTp.data(), Zp.data(), n, p0 + KB_NB, n);
            for (int c = p0 + KB_NB; c < n; ++c)
                for (int r = 0; r < n; ++r) {
                    cd acc = W[r + (size_t)c * n];
                    for (int t = 0; t < KB_NB; ++t)
                        acc = acc - Yp[r + (size_t)t * n] * conj(hess_vt(W.data(), n, p0, c, t)) -
                              hess_vt(W.data(), n, p0, r, t) * conj(Zp[c + (size_t)t * n]);
                    W[r + (size_t)c * n] = acc;
                }
        }
        gehd2(ctx, n, W.data(), n, th.data(), npan * KB_NB);
    }
    gen_unitary_cols<HostCtx, HS_MAXC>(ctx, n, n - 2, 1, W.data(), n, th.data(), Qh.data(), n, 0, n);
    hess_copy(ctx, n, W.data(), n, Hc.data(), n);
    // infinity norm of H (zhsein: hnorm = zlanhs('I'))
    double hnorm = 0.0;
    for (int i = 0; i < n; ++i) {
        double r = 0.0;
        for (int j = 0; j < n; ++j) r += cabs(Hc[i + (size_t)j * n]);
        hnorm = r > hnorm ? r : hnorm;
    }
    int info = 0, weak = 0;
    cd* mu = reinterpret_cast<cd*>(mu_out);
    // the device's order: the Ehrlich-Aberth path, the QR iteration (what k_hqr2 runs) for a member it declines
    const char* abz = getenv("HS_EIG_AB");
    if ((abz && atoi(abz) == 0) || hs_aberth_path(ctx, n, Hc.data(), hnorm, mu, nullptr) != 0)
        hqr2_eigvals(ctx, n, Hc.data(), n, mu, &info, KB2_NBMAX, KB2_WIN_DEV);
    invit<HostCtx, 4096>(ctx, n, W.data(), n, mu, hnorm, X.data(), n, 1, &weak);
    // P = Qh * X
    cd* P = reinterpret_cast<cd*>(P_out);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            cd acc = czero();
            for (int k = 0; k < n; ++k) cfma(acc, Qh[i + (size_t)k * n], X[k + (size_t)j * n]);
            P[i + (size_t)j * n] = acc;
        }
    return info | (weak ? 4 : 0);
}


// The divide-and-conquer Ehrlich-Aberth path on its own: W (n x n column-major, any matrix: reduced to Hessenberg form
// first) -> mu; out[0] = 1 if the path declined (the caller would run the QR iteration), out[1] = root iterations.
int hs_eig_aberth(const double* W_in, int n, double* mu_out, long long* out) {
    std::vector<cd> W(n * n), Hc(n * n), th(n);
    memcpy(W.data(), W_in, sizeof(cd) * n * n);
    std::vector<char> arena;
    HostCtx ctx = make_ctx(arena, gehd2_scratch_bytes(n, 1, 1) + 4096);
    gehd2(ctx, n, W.data(), n, th.data());
    hess_copy(ctx, n, W.data(), n, Hc.data(), n);
    double hnorm = 0.0;
    for (int i = 0; i < n; ++i) {
        double r = 0.0;
        for (int j = 0; j < n; ++j) r += cabs(Hc[i + (size_t)j * n]);
        hnorm = r > hnorm ? r : hnorm;
    }
    long long iters = 0;
    const int declined = hs_aberth_path(ctx, n, Hc.data(), hnorm, reinterpret_cast<cd*>(mu_out), &iters);
    out[0] = declined; out[1] = iters;
    return 0;
}

// The Aberth / Hyman small eigenvalue solver on its own: T upper Hessenberg n x n (column-major), z out.
// Returns the solver's verdict (1 = converged).
int hs_aberth(const double* T_in, int n, double* z_out) {
    std::vector<cd> T(n * n), U(n * n), D(n * n), zw(2 * n), z(n);
    memcpy(T.data(), T_in, sizeof(cd) * n * n);
    std::vector<char> arena;
    HostCtx ctx = make_ctx(arena, 1024);
    WaveCtx<HostCtx> wc{ctx, nullptr, 0};
    std::vector<cd> zr(3 * n);
    const bool ok = (n == 8) ? aberth_eigs_reg<8>(wc, T.data(), n, z.data(), zr.data(), 40, nullptr)
                             : aberth_eigs(wc, n, T.data(), n, z.data(), U.data(), D.data(), zw.data(), 40);
    memcpy(z_out, z.data(), sizeof(cd) * n);
    return ok ? 1 : 0;
}

// Debug trace of the second-generation iteration: one line per window step (geometry + a checksum of H) into the
// file named by HS_TRACE.
static FILE* g_trace = nullptr;
static const cd* g_trace_H = nullptr;
static int g_trace_n = 0;
static void hs_trace_step(const Win2Geom& G, int team, int) {
    if (!g_trace) return;
    double s1 = 0, s2 = 0;
    for (size_t k = 0; k < (size_t)g_trace_n * g_trace_n; ++k) { s1 += g_trace_H[k].x * (1 + (k % 7)); s2 += g_trace_H[k].y * (1 + (k % 5)); }
    fprintf(g_trace, "l %d i %d nb %d t0 %d t1 %d ws %d we %d bmin %d bmax %d near %d team %d sum %.17g %.17g\n", G.l, G.i, G.nb, G.t0, G.t1,
            G.ws, G.we, G.bmin, G.bmax, G.nr_near, team, s1, s2);
}

// Second-generation QR iteration (kb_hqr2.hpp): double-shift bulges, time-major log, strip units.
// team != 0 runs the chase-workgroup / helper-workgroup split with the helper's share inline.
int hs_eigvals2(const double* W_in, int n, int nbmax, int win_w, int team, double* mu_out, long long* stats_out, int smode) {
    std::vector<cd> W(n * n), Hc(n * n), th(n);
    memcpy(W.data(), W_in, sizeof(cd) * n * n);
    std::vector<char> arena;
    HostCtx ctx = make_ctx(arena, gehd2_scratch_bytes(n, 1, 1) + hqr2_scratch_bytes(win_w));
    gehd2(ctx, n, W.data(), n, th.data());
    hess_copy(ctx, n, W.data(), n, Hc.data(), n);
    int info = 0;
    MsStats st;
    memset(&st, 0, sizeof(st));
    TeamCtl ctl;
    memset(&ctl, 0, sizeof(ctl));
    Team2<HostCtx> tm;
    tm.ctl = &ctl;
    tm.rec_bytes = team2_rec_bytes(win_w);
    std::vector<char> ring((size_t)KB_TEAM_SLOTS * tm.rec_bytes);
    tm.ring = ring.data();
    tm.g = 0; tm.g_batch = 0; tm.failed = 0;
    tm.A = HSc1::make(Hc.data(), n, n);
    tm.W = win_w;
    if (const char* tf = getenv("HS_TRACE")) {
        g_trace = fopen(tf, "w"); g_trace_H = Hc.data(); g_trace_n = n; kb2_host_trace = hs_trace_step;
    }
    hqr2_eigvals(ctx, n, Hc.data(), n, reinterpret_cast<cd*>(mu_out), &info, nbmax, win_w, &st, team ? &tm : nullptr, smode);
    if (g_trace) { fclose(g_trace); g_trace = nullptr; kb2_host_trace = nullptr; }
    if (stats_out) {
        stats_out[0] = st.intervals; stats_out[1] = st.batches; stats_out[2] = st.single_sweeps; stats_out[3] = st.small_steps;
        stats_out[4] = ctl.published; stats_out[5] = ctl.all_done; stats_out[6] = ctl.done; stats_out[7] = ctl.near_done;
        stats_out[8] = st.ab_calls; stats_out[9] = st.ab_fail; stats_out[10] = st.ab_iters;
    }
    return info;
}

// Full single-member pipeline (reference kbdm.py:19-92) with the host context.
//   signal: N complex; lines: l x 4 row-major (A, T2, F, PH); sv: m; mu: l complex
int hs_kbdm(const double* signal, int N, int m, int l, int p, double q, double dwell,
            double* lines, double* sv, double* mu_out) {
    const cd* c = reinterpret_cast<const cd*>(signal);
    (void)N;
    std::vector<double> A((size_t)2 * m * m), L((size_t)2 * m * m), R((size_t)2 * m * m), s(m);
    cd* Ac = reinterpret_cast<cd*>(A.data());
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < m; ++i) Ac[i + (size_t)j * m] = c[i + j + p - 1];   // U^{p-1}
    int info = hs_svd(A.data(), m, L.data(), s.data(), R.data());
    for (int i = 0; i < m; ++i) sv[i] = s[i];
    const cd* Lc = reinterpret_cast<const cd*>(L.data());
    const cd* Rc = reinterpret_cast<const cd*>(R.data());
    std::vector<double> dsqi(l);
    for (int i = 0; i < l; ++i) dsqi[i] = (q > 0) ? 1.0 / sqrt(s[i] + q * q / s[i]) : 1.0 / sqrt(s[i]);
    // T1 = Up R_  (m x l),  W = Dsqi L_^H T1 Dsqi (l x l)
    std::vector<cd> T1((size_t)m * l), W((size_t)l * l);
    for (int j = 0; j < l; ++j)
        for (int i = 0; i < m; ++i) {
            cd acc = czero();
            for (int k = 0; k < m; ++k) cfma(acc, c[i + k + p], Rc[k + (size_t)j * m]);
            T1[i + (size_t)j * m] = acc;
        }
    for (int j = 0; j < l; ++j)
        for (int i = 0; i < l; ++i) {
            cd acc = czero();
            for (int k = 0; k < m; ++k) cfmac(acc, Lc[k + (size_t)i * m], T1[k + (size_t)j * m]);
            W[i + (size_t)j * l] = (dsqi[i] * dsqi[j]) * acc;
        }
    std::vector<double> P((size_t)2 * l * l);
    info |= hs_eig(reinterpret_cast<double*>(W.data()), l, mu_out, P.data());
    const cd* Pc = reinterpret_cast<const cd*>(P.data());
    const cd* mu = reinterpret_cast<const cd*>(mu_out);
    // B = R_ Dsqi P (m x l); N_k = b_k^T U0 b_k ; dsq_k = c[:m] . b_k
    std::vector<cd> B((size_t)m * l);
    for (int k = 0; k < l; ++k)
        for (int i = 0; i < m; ++i) {
            cd acc = czero();
            for (int j = 0; j < l; ++j) cfma(acc, Rc[i + (size_t)j * m], dsqi[j] * Pc[j + (size_t)k * l]);
            B[i + (size_t)k * m] = acc;
        }
    for (int k = 0; k < l; ++k) {
        cd nk = czero(), ds = czero();
        for (int i = 0; i < m; ++i) {
            cd t = czero();
            for (int j = 0; j < m; ++j) cfma(t, c[i + j], B[j + (size_t)k * m]);
            cfma(nk, B[i + (size_t)k * m], t);
            cfma(ds, c[i], B[i + (size_t)k * m]);
        }
        const cd D = cdiv(ds * ds, nk);
        const double lnabs = log(cabs(mu[k]));
        lines[4 * k + 0] = cabs(D);
        lines[4 * k + 1] = -dwell / lnabs;
        lines[4 * k + 2] = atan2(mu[k].y, mu[k].x) / (2.0 * M_PI * dwell);
        lines[4 * k + 3] = atan2(D.y, D.x);
    }
    return info;
}

}  // extern "C"
