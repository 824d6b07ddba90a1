// Panels of the two blocked reductions for a team of T >= 1 workgroups per member (kb_team.hpp).  Same algorithms and the
// same panel factors as kb_svd.hpp: bidiag_panel / kb_eig.hpp: hess_panel; what changes is the organisation:
//   * the current column lives in LDS (every workgroup of the team holds all of it), the Householder generators run there;
//   * the products with the untouched trailing matrix are cut into slots (kb_team.hpp) owned by the workgroups;
//   * what a slot owner computes for its columns / rows travels through the team's exchange buffer, after which EVERY
//     workgroup writes ALL of the panel arrays V, X, Y, U (the same values): later reads hit the workgroup's own stores.
// The arithmetic of every output element is the same whatever T is.
#pragma once
#include "kb_eig.hpp"
#include "kb_svd.hpp"
#include "kb_team.hpp"

// Phase timers of the panels (diagnostic builds only: -DKB_PANEL_PROF; tools/panel_phases.py): thread 0 of role 0 adds the
// 100 MHz wall-clock ticks between marks to kb_panel_prof[phase].
#if defined(KB_PANEL_PROF) && defined(__HIPCC__)
__device__ unsigned long long kb_panel_prof[32];
#endif
#if defined(KB_PANEL_PROF) && defined(__HIP_DEVICE_COMPILE__)
#define KB_PROF_DECL unsigned long long prof_t = wall_clock64(); const bool prof_on = ctx.tid() == 0 && tm.role == 0;
#define KB_PROF(ph_) do { if (prof_on) { const unsigned long long t_ = wall_clock64(); atomicAdd(&kb_panel_prof[ph_], t_ - prof_t); prof_t = t_; } } while (0)
#define KB_PROF_SYNC(ph_) do { ctx.sync(); KB_PROF(ph_); } while (0)      /* (a barrier the production build does not have) */
#else
#define KB_PROF_DECL
#define KB_PROF(ph_) do { } while (0)
#define KB_PROF_SYNC(ph_) do { } while (0)
#endif


namespace kb {

// LDS: column (n) + row (n) + raw dots / y_j (n) + x_{j-1} (n) + sweep results (2 n) + row-product partial sums (NCG x zr)
// + small vectors
KB_HD int panel_team_scratch_bytes(int n, int zr, int nthreads) {
    (void)nthreads;
    return (6 * n + KB_TEAM_NCG * zr + 8 * KB_NB + KB_NB * KB_NB + 8) * (int)sizeof(cd);
}
// exchange buffer entries per member (bidiagonalisation: y, raw row, x, next column; Hessenberg: y, next column)
KB_HD int panel_team_xbuf_elems(int n) { return 4 * n; }

// Row product  h(i) = sum_c Ab[i, c] u[c]  (i < nr local rows, c < ncols; lanes along rows) for the row chunks this
// workgroup owns (chunk rc = role + k T, rows rc WS .. rc WS + WS - 1), chunks [k0, k1) of them in this batch:
// partial sums of column group cg (columns cg, cg + NCG, ...) go to zp[cg * zr + (k - k0) * WS + lane].
template <class C>
KB_HD void team_row_product(const C& ctx, const PanelTeam<C>& tm, int nr, int ncols, const cd* __restrict__ Ab, int ld,
                            const cd* u, cd* zp, int zr, int k0, int k1) {
    const int lane = ctx.lane();
    const int nslots = (k1 - k0) * KB_TEAM_NCG;
    for (int slot = ctx.wave(); slot < nslots; slot += ctx.nwaves()) {
        const int kk = slot / KB_TEAM_NCG, cg = slot - kk * KB_TEAM_NCG;
        const int rc = tm.role + (k0 + kk) * tm.T;
        const int i = rc * C::WS + lane;
        if (i < nr) {
            cd acc = czero(), acc1 = czero(), acc2 = czero(), acc3 = czero();
            int c = cg;
            for (; c + 7 * KB_TEAM_NCG < ncols; c += 8 * KB_TEAM_NCG) {       // eight independent loads in flight
                const cd* __restrict__ ap = Ab + i + (size_t)c * ld;
                const size_t st = (size_t)KB_TEAM_NCG * ld;
                const cd a0 = ap[0], a1 = ap[st], a2 = ap[2 * st], a3 = ap[3 * st];
                const cd a4 = ap[4 * st], a5 = ap[5 * st], a6 = ap[6 * st], a7 = ap[7 * st];
                cfma(acc, a0, u[c]); cfma(acc1, a1, u[c + KB_TEAM_NCG]);
                cfma(acc2, a2, u[c + 2 * KB_TEAM_NCG]); cfma(acc3, a3, u[c + 3 * KB_TEAM_NCG]);
                cfma(acc, a4, u[c + 4 * KB_TEAM_NCG]); cfma(acc1, a5, u[c + 5 * KB_TEAM_NCG]);
                cfma(acc2, a6, u[c + 6 * KB_TEAM_NCG]); cfma(acc3, a7, u[c + 7 * KB_TEAM_NCG]);
            }
            for (; c + 3 * KB_TEAM_NCG < ncols; c += 4 * KB_TEAM_NCG) {       // four
                const cd a0 = Ab[i + (size_t)c * ld], a1 = Ab[i + (size_t)(c + KB_TEAM_NCG) * ld];
                const cd a2 = Ab[i + (size_t)(c + 2 * KB_TEAM_NCG) * ld], a3 = Ab[i + (size_t)(c + 3 * KB_TEAM_NCG) * ld];
                cfma(acc, a0, u[c]); cfma(acc1, a1, u[c + KB_TEAM_NCG]);
                cfma(acc2, a2, u[c + 2 * KB_TEAM_NCG]); cfma(acc3, a3, u[c + 3 * KB_TEAM_NCG]);
            }
            if (c < ncols) {                                                   // up to three columns left: their loads together
                const int c1 = c + KB_TEAM_NCG, c2 = c + 2 * KB_TEAM_NCG;
                const bool h1 = c1 < ncols, h2 = c2 < ncols;
                const cd a0 = Ab[i + (size_t)c * ld];
                const cd a1 = h1 ? Ab[i + (size_t)c1 * ld] : czero();
                const cd a2 = h2 ? Ab[i + (size_t)c2 * ld] : czero();
                cfma(acc, a0, u[c]);
                if (h1) cfma(acc1, a1, u[c1]);
                if (h2) cfma(acc2, a2, u[c2]);
            }
            zp[cg * zr + kk * C::WS + lane] = (acc + acc1) + (acc2 + acc3);
        }
    }
}
// conj(a(i)) . b(i) over i = i0 + lane, i0 + lane + WS, ... < n with four loads in flight (a: memory, b: LDS); the caller
// reduces over the lanes.  Terms are added chunk by chunk into four accumulators, combined as (s0 + s1) + (s2 + s3).
template <class C, class FA>
KB_HD cd team_dotc4(const C& ctx, int i0, int n, FA a, const cd* b) {
    cd s0 = czero(), s1 = czero(), s2 = czero(), s3 = czero();
    int i = i0 + ctx.lane();
    for (; i + 3 * C::WS < n; i += 4 * C::WS) {
        const cd a0 = a(i), a1 = a(i + C::WS), a2 = a(i + 2 * C::WS), a3 = a(i + 3 * C::WS);
        cfmac(s0, a0, b[i]); cfmac(s1, a1, b[i + C::WS]); cfmac(s2, a2, b[i + 2 * C::WS]); cfmac(s3, a3, b[i + 3 * C::WS]);
    }
    if (i + 2 * C::WS < n) {
        const cd a0 = a(i), a1 = a(i + C::WS), a2 = a(i + 2 * C::WS);
        cfmac(s0, a0, b[i]); cfmac(s1, a1, b[i + C::WS]); cfmac(s2, a2, b[i + 2 * C::WS]);
    } else if (i + C::WS < n) {
        const cd a0 = a(i), a1 = a(i + C::WS);
        cfmac(s0, a0, b[i]); cfmac(s1, a1, b[i + C::WS]);
    } else if (i < n) {
        cfmac(s0, a(i), b[i]);
    }
    return (s0 + s1) + (s2 + s3);
}

KB_HD cd team_row_sum(const cd* zp, int zr, int off) {
    cd h = czero();
    for (int cg = 0; cg < KB_TEAM_NCG; ++cg) h += zp[cg * zr + off];
    return h;
}

// A sweep over the panel factors: for each of nq entries (a row or a column of the panel arrays) two sums over the
// reflectors t so far.  KB_SWEEP_G = 4 work items per entry take t = g, g + 4, ... (g < 4); their partial sums meet in the
// fixed tree (p0 + p1) + (p2 + p3) - on the device four adjacent lanes and two DPP steps, on the host the same tree spelled
// out.  part(q, g, a, b) accumulates into a, b; out(q, a, b) receives the sums (one call per entry).
constexpr int KB_SWEEP_G = 4;
KB_HD cd tree4(const cd* p) { return (p[0] + p[1]) + (p[2] + p[3]); }
template <class C, class FP, class FO>
KB_HD void team_sweep8(const C& ctx, int nq, FP part, FO out) {
    if (C::QUAD == 1) {
        for (int q = ctx.tid(); q < nq; q += ctx.nthreads()) {
            cd a[KB_SWEEP_G], b[KB_SWEEP_G];
            for (int g = 0; g < KB_SWEEP_G; ++g) { a[g] = czero(); b[g] = czero(); part(q, g, a[g], b[g]); }
            out(q, tree4(a), tree4(b));
        }
    } else {
        const int nitems = nq * KB_SWEEP_G, nt = ctx.nthreads();
        for (int e0 = 0; e0 < nitems; e0 += nt) {               // (uniform trip count: the DPP steps need whole wavefronts)
            const int e = e0 + ctx.tid();
            const bool act = e < nitems;
            const int q = act ? (e / KB_SWEEP_G) : 0, g = e % KB_SWEEP_G;
            cd a = czero(), b = czero();
            if (act) part(q, g, a, b);
            a = mk(ctx.quad_sum(a.x), ctx.quad_sum(a.y));
            b = mk(ctx.quad_sum(b.x), ctx.quad_sum(b.y));
            if (act && g == 0) out(q, a, b);
        }
    }
}

// Column dots  yr[c] = A0[j:, c]^H v  for the blocks of KB_TEAM_CB columns this workgroup owns (block b = role + kb T:
// columns j + 1 + b CB ...): one wavefront per pair of columns, lanes along the rows, eight loads in flight.
template <class C>
KB_HD void team_col_dots(const C& ctx, int role, int T, int nownb, int n, int j, const cd* __restrict__ A, int ld,
                                     const cd* vc, cd* yr) {
#define A_(r_, c_) A[(r_) + (size_t)(c_) * ld]
    const int lane = ctx.lane(), nw = ctx.nwaves();
    for (int pp = ctx.wave(); pp < nownb * (KB_TEAM_CB / 2); pp += nw) {
        const int kb = pp / (KB_TEAM_CB / 2), pr = pp - kb * (KB_TEAM_CB / 2);
        const int c0 = j + 1 + (role + kb * T) * KB_TEAM_CB + 2 * pr;
        if (c0 >= n) continue;
        const int c1 = (c0 + 1 < n) ? c0 + 1 : c0;
        cd g0 = czero(), g1 = czero(), h0 = czero(), h1 = czero();
        int r = j + lane;
        for (; r + C::WS < n; r += 2 * C::WS) {
            const cd a00 = A_(r, c0), a01 = A_(r, c1), a10 = A_(r + C::WS, c0), a11 = A_(r + C::WS, c1);
            const cd v0 = vc[r], v1 = vc[r + C::WS];
            cfmac(g0, a00, v0); cfmac(g1, a01, v0); cfmac(h0, a10, v1); cfmac(h1, a11, v1);
        }
        for (; r < n; r += C::WS) {
            const cd a00 = A_(r, c0), a01 = A_(r, c1);
            const cd v0 = vc[r];
            cfmac(g0, a00, v0); cfmac(g1, a01, v0);
        }
        const cd s0 = ctx.wave_sum(g0 + h0);
        const cd s1 = ctx.wave_sum(g1 + h1);
        if (lane == 0) {
            yr[c0] = s0;
            if (c0 + 1 < n) yr[c0 + 1] = s1;
        }
    }
#undef A_
}

// ---------------------------------------------------------------------------------
// Blocked bidiagonalisation panel (the derivation is at kb_svd.hpp: bidiag_panel).  zr: rows of the partial-sum array
// (a multiple of the wave size).  Returns false when the team gave up.
// Barriers: the panel arrays V, X, Y, U in memory are stored and forgotten (sync_lds does not wait for stores); a phase
// reads them only from columns whose stores lie behind a synchronisation of the team (which drains them), the two vectors
// that are needed sooner - y_j, x_{j-1} - are kept in LDS as well.
template <class C>
KB_HD bool bidiag_panel_team(const C& ctx, PanelTeam<C>& tm, int n, cd* A, int ld, double* d, double* e, cd* tauq,
                             cd* taup, cd* UR, int ldr, cd* X, cd* Y, int ldxy, int zr) {
#define A_(r_, c_) A[(r_) + (size_t)(c_) * ld]
#define U_(r_, c_) UR[(r_) + (size_t)(c_) * ldr]
#define X_(r_, c_) X[(r_) + (size_t)(c_) * ldxy]
#define Y_(r_, c_) Y[(r_) + (size_t)(c_) * ldxy]
    const int tid = ctx.tid(), nt = ctx.nthreads(), lane = ctx.lane(), nw = ctx.nwaves();
    const int T = tm.T, role = tm.role;
    cd* vc = reinterpret_cast<cd*>(ctx.scratch());           // current column (indexed by row): v_j after the generator
    cd* ub = vc + n;                                          // raw row -> u_j (indexed by c - j - 1)
    cd* yr = ub + n;                                          // raw column dots of the owned columns, then y_j (indexed by c)
    cd* xl = yr + n;                                          // x_{j-1} (indexed by r)
    cd* sp = xl + n;                                          // results of the sweep over the rows: sp[r], sp[n + r]
    cd* zp = sp + 2 * n;                                      // partial sums of the row product: NCG x zr
    cd* w1 = zp + KB_TEAM_NCG * zr;
    cd* w2 = w1 + KB_NB;
    cd* z1 = w2 + KB_NB;
    cd* z2 = z1 + KB_NB;
    cd* ra = z2 + KB_NB;                                      // row j of V, row j of X; later row j+1 of Y, of U
    cd* rb = ra + KB_NB;
    const int zrc = zr / C::WS;                               // row chunks per batch of the row product
    const bool writer = role == 0;
    KB_PROF_DECL
    for (int r = tid; r < n; r += nt) vc[r] = A_(r, 0);
    if (!tm.sync(ctx)) return false;       // (nobody stores v_0 over the raw column before every workgroup has read it)
    KB_PROF(0);
    for (int j = 0; j < KB_NB; ++j) {
        // ---- left reflector from the column in LDS; v to the panel (every workgroup, the same values)
        double beta;
        cd tq;
        // rows j of V and of X for the sweep below, fetched under the generator (X(:, j-1) from LDS: its stores are not behind
        // a synchronisation yet); KB_NB <= the workgroup size on the device, the host loops
        cd pra = czero(), prb = czero();
        if (C::WS > 1 && tid < j) { pra = A_(j, tid); prb = (tid == j - 1) ? xl[j] : X_(j, tid); }
        // w1 = V^H v, w2 = X^H v (replicated): v = [1; s x] with the generator's scale s, so the dots with the RAW tail x run
        // in front of the generator's barrier (larfg_front) and w = conj(row j) + s (tail^H x) follows without another pass
        const cd sl = larfg_front(ctx, n - j, vc + j, beta, tq, [&]() {
            for (int t = ctx.wave(); t < 2 * j; t += nw) {
                const bool second = t >= j;
                const int tt = second ? t - j : t;
                cd acc;
                if (second && tt == j - 1) acc = team_dotc4(ctx, j + 1, n, [&](int r) { return xl[r]; }, vc);
                else {
                    const cd* __restrict__ src = second ? &X_(0, tt) : &A_(0, tt);
                    acc = team_dotc4(ctx, j + 1, n, [&](int r) { return src[r]; }, vc);
                }
                acc = ctx.wave_sum(acc);
                if (lane == 0) { if (second) w2[tt] = acc; else w1[tt] = acc; }
            }
        });
        if (tid == 0) {
            vc[j] = mk(1.0, 0.0);
            if (writer) { d[j] = beta; tauq[j] = tq; }
        }
        for (int r = j + 1 + tid; r < n; r += nt) A_(r, j) = vc[r];             // (the entries this thread has just scaled)
        if (C::WS > 1) {
            if (tid < j) {
                ra[tid] = pra; rb[tid] = prb;
                w1[tid] = conj(pra) + w1[tid] * sl;
                w2[tid] = conj(prb) + w2[tid] * sl;
            }
        } else
            for (int t = tid; t < j; t += nt) {
                ra[t] = A_(j, t); rb[t] = (t == j - 1) ? xl[j] : X_(j, t);
                w1[t] = conj(ra[t]) + w1[t] * sl;
                w2[t] = conj(rb[t]) + w2[t] * sl;
            }
        ctx.sync_lds();
        KB_PROF(1);
        KB_PROF(2);
        const int nr = n - j - 1;
        const int nblk = (nr + KB_TEAM_CB - 1) / KB_TEAM_CB;
        const int nownb = team_own_count(nblk, role, T);
        team_col_dots(ctx, role, T, nownb, n, j, A, ld, vc, yr);
        ctx.sync_lds();
        KB_PROF(3);
        // ---- row j of H_j^H A^(j), columns j+1.., conjugated, and y_c: the owned columns
        team_sweep8(ctx, nownb * KB_TEAM_CB,
            [&](int q, int g, cd& acc, cd& co) {
                const int c = j + 1 + (role + (q / KB_TEAM_CB) * T) * KB_TEAM_CB + (q % KB_TEAM_CB);
                if (c >= n) return;
                int t = g;
                for (; t + KB_SWEEP_G < j; t += 2 * KB_SWEEP_G) {              // two reflectors per step: four loads in flight
                    const int t2 = t + KB_SWEEP_G;
                    const cd yct = Y_(c, t), uct = U_(c, t), yc2 = Y_(c, t2), uc2 = U_(c, t2);
                    acc = acc + ra[t] * conj(yct) + rb[t] * conj(uct);
                    co = co + yct * w1[t] + uct * w2[t];
                    acc = acc + ra[t2] * conj(yc2) + rb[t2] * conj(uc2);
                    co = co + yc2 * w1[t2] + uc2 * w2[t2];
                }
                if (t < j) {
                    const cd yct = Y_(c, t), uct = U_(c, t);
                    acc = acc + ra[t] * conj(yct) + rb[t] * conj(uct);
                    co = co + yct * w1[t] + uct * w2[t];
                }
            },
            [&](int q, cd acc, cd co) {
                const int c = j + 1 + (role + (q / KB_TEAM_CB) * T) * KB_TEAM_CB + (q % KB_TEAM_CB);
                if (c >= n) return;
                const cd yc = tq * (yr[c] - co);               // y_c = tauq (A0[:, c]^H v - Y[c, :j] w1 - U[c, :j] w2)
                const cd rr = conj((A_(j, c) - acc) - conj(yc));
                if (T == 1) {                                  // a team of one owns every column: straight to where they go
                    Y_(c, j) = yc;
                    yr[c] = yc;                                // (this thread alone read the raw dot there)
                    ub[c - j - 1] = rr;
                } else {
                    tm.put(c, yc);
                    tm.put(n + c, rr);
                }
            });
        KB_PROF(4);
        if (!tm.sync(ctx)) return false;
        KB_PROF(5);
        if (T > 1)
            for (int c = j + 1 + tid; c < n; c += nt) {
                const cd yc = tm.get(c);
                Y_(c, j) = yc;
                yr[c] = yc;
                ub[c - j - 1] = tm.get(n + c);
            }
        for (int c = tid; c <= j; c += nt) Y_(c, j) = czero();
        ctx.sync_lds();
        KB_PROF(6);
        // ---- right reflector (replicated)
        double be;
        cd tp;
        const bool nextcol = j + 1 < KB_NB;
        // rows j+1 of Y and of U (for the dots below and for the sweep), fetched under the generator (Y(:, j) from LDS)
        if (C::WS > 1 && tid <= j) { pra = (tid == j) ? yr[j + 1] : Y_(j + 1, tid); prb = (tid < j) ? U_(j + 1, tid) : czero(); }
        // ---- x_j = taup (A^(j) u - v (y^H u)), rows j+1..n-1: z1 = Y^H u (t <= j), z2 = U^H u (t < j) replicated, from the
        // raw row in front of the generator's barrier as on the left ...
        const cd su = larfg_front(ctx, nr, ub, be, tp, [&]() {
            const cd* ubc = ub - (j + 1);                          // ubc[c] = u_j(c)
            for (int t = ctx.wave(); t < 2 * j + 1; t += nw) {
                const bool second = t > j;
                const int tt = second ? t - j - 1 : t;
                cd acc;
                if (!second && tt == j) acc = team_dotc4(ctx, j + 2, n, [&](int c) { return yr[c]; }, ubc);
                else {
                    const cd* __restrict__ src = second ? &U_(0, tt) : &Y_(0, tt);
                    acc = team_dotc4(ctx, j + 2, n, [&](int c) { return src[c]; }, ubc);
                }
                acc = ctx.wave_sum(acc);
                if (lane == 0) { if (second) z2[tt] = acc; else z1[tt] = acc; }
            }
        });
        if (tid == 0) {
            ub[0] = mk(1.0, 0.0);
            if (writer) { e[j] = be; taup[j] = tp; }
        }
        if (C::WS > 1) {
            if (tid <= j) {
                ra[tid] = pra; rb[tid] = prb;
                z1[tid] = conj(pra) + z1[tid] * su;
                if (tid < j) z2[tid] = conj(prb) + z2[tid] * su;
            }
        } else
            for (int t = tid; t <= j; t += nt) {
                ra[t] = (t == j) ? yr[j + 1] : Y_(j + 1, t); rb[t] = (t < j) ? U_(j + 1, t) : czero();
                z1[t] = conj(ra[t]) + z1[t] * su;
                if (t < j) z2[t] = conj(rb[t]) + z2[t] * su;
            }
        ctx.sync_lds();
        for (int c = j + 1 + tid; c < n; c += nt) U_(c, j) = ub[c - j - 1];     // explicit 1 at row j+1
        KB_PROF(7);
        KB_PROF(8);
        // ---- ... the corrections of x_j with the panel factors and column j + 1 of A^(j+1), for the owned rows
        const int nrc = (nr + C::WS - 1) / C::WS;
        const int nownc = team_own_count(nrc, role, T);
        team_sweep8(ctx, nownc * C::WS,
            [&](int q, int g, cd& hp, cd& np) {
                const int i = (role + (q / C::WS) * T) * C::WS + (q % C::WS);
                if (i >= nr) return;
                const int r = j + 1 + i;
                int t = g;
                for (; t + KB_SWEEP_G < j; t += 2 * KB_SWEEP_G) {              // (both reflectors below j: X exists for them)
                    const int t2 = t + KB_SWEEP_G;
                    const cd art = A_(r, t), xrt = X_(r, t), ar2 = A_(r, t2), xr2 = X_(r, t2);
                    hp = hp + art * z1[t];
                    if (nextcol) np = np + art * conj(ra[t]);
                    hp = hp + xrt * z2[t];
                    if (nextcol) np = np + xrt * conj(rb[t]);
                    hp = hp + ar2 * z1[t2];
                    if (nextcol) np = np + ar2 * conj(ra[t2]);
                    hp = hp + xr2 * z2[t2];
                    if (nextcol) np = np + xr2 * conj(rb[t2]);
                }
                for (; t <= j; t += KB_SWEEP_G) {
                    const cd art = A_(r, t);
                    hp = hp + art * z1[t];
                    if (nextcol) np = np + art * conj(ra[t]);
                    if (t < j) {
                        const cd xrt = X_(r, t);
                        hp = hp + xrt * z2[t];
                        if (nextcol) np = np + xrt * conj(rb[t]);
                    }
                }
            },
            [&](int q, cd hp, cd np) {
                const int i = (role + (q / C::WS) * T) * C::WS + (q % C::WS);
                if (i >= nr) return;
                sp[j + 1 + i] = hp;
                sp[n + j + 1 + i] = np;
            });
        // ---- ... and h = A0[j+1:, j+1:] u for the owned chunks of rows, in batches of zrc chunks
        KB_PROF_SYNC(9);
        for (int k0 = 0; k0 < nownc; k0 += zrc) {
            const int k1 = (k0 + zrc < nownc) ? k0 + zrc : nownc;
            team_row_product(ctx, tm, nr, nr, &A_(j + 1, j + 1), ld, ub, zp, zr, k0, k1);
            ctx.sync_lds();
            KB_PROF(10);
            for (int q = tid; q < (k1 - k0) * C::WS; q += nt) {
                const int i = (role + (k0 + q / C::WS) * T) * C::WS + (q % C::WS);
                if (i >= nr) continue;
                const int r = j + 1 + i;
                const cd h = team_row_sum(zp, zr, q);
                const cd xr = tp * (h - sp[r]);
                const cd nx = nextcol ? (A_(r, j + 1) - sp[n + r]) - xr : czero();   // ... - X(r, j) conj(U(j+1, j)), U(j+1, j) = 1
                if (T == 1) {
                    X_(r, j) = xr;
                    xl[r] = xr;
                    if (nextcol) vc[r] = nx;
                } else {
                    tm.put(2 * n + r, xr);
                    if (nextcol) tm.put(3 * n + r, nx);
                }
            }
            ctx.sync_lds();
            KB_PROF(11);
        }
        if (!tm.sync(ctx)) return false;
        KB_PROF(12);
        if (T > 1)
            for (int r = j + 1 + tid; r < n; r += nt) {
                const cd xr = tm.get(2 * n + r);
                X_(r, j) = xr;
                xl[r] = xr;
                if (nextcol) vc[r] = tm.get(3 * n + r);
            }
        for (int r = tid; r <= j; r += nt) { X_(r, j) = czero(); xl[r] = czero(); }
        ctx.sync_lds();
        KB_PROF(13);
    }
    return true;
#undef A_
#undef U_
#undef X_
#undef Y_
}

// ---------------------------------------------------------------------------------
// Blocked Hessenberg panel (the derivation is at kb_eig.hpp: hess_panel).  One synchronisation of the team per column.
// Rows r > p0 only (zlahr2's organisation): the rows above the panel take no part in its reflectors - their share of Y and
// their panel columns follow in ONE product after the panel (kb_eig.hpp: hess_ytop_block / k_hess_ytop), so the pass over
// A0 of column k reads (N - p0 - 1) x (N - k - 1) entries instead of N x (N - k - 1).
// (Barriers as in bidiag_panel_team; the vector that is needed before its stores are drained is y_{j-1}: LDS.)
template <class C>
KB_HD bool hess_panel_team(const C& ctx, PanelTeam<C>& tm, int N, cd* W, int ld, int p0, cd* tauh, cd* Y, int ldy,
                           cd* VT, int ldvt, cd* MT, int zr) {
#define W_(r_, c_) W[(r_) + (size_t)(c_) * ld]
#define Y_(r_, c_) Y[(r_) + (size_t)(c_) * ldy]
#define T_(r_, c_) Tm[(r_) + (c_) * KB_NB]
    const int tid = ctx.tid(), nt = ctx.nthreads(), lane = ctx.lane(), nw = ctx.nwaves();
    const int T = tm.T, role = tm.role;
    cd* xc = reinterpret_cast<cd*>(ctx.scratch());           // current column (all N rows), then v (rows k+1..)
    cd* yl = xc + N;                                          // y_{j-1} (all rows)
    cd* sp = yl + N;                                          // results of the sweeps: 2 N
    cd* yn = sp + 2 * N;                                      // a team of one: y_j and the next column on their way (2 N)
    cd* xn = yn + N;
    cd* zp = xc + 6 * N;                                      // (the layout of panel_team_scratch_bytes)
    cd* w1 = zp + KB_TEAM_NCG * zr;
    cd* w2 = w1 + KB_NB;
    cd* ra = w2 + KB_NB;                                      // row k+1 of V
    cd* Tm = w1 + 8 * KB_NB;                                  // T, NB x NB upper triangular
    const int zrc = zr / C::WS;
    const bool writer = role == 0;
    const int R0 = p0 + 1, NR = N - R0;                       // the panel's rows
    const int nrc = (NR + C::WS - 1) / C::WS;
    const int nownc = team_own_count(nrc, role, T);
    for (int idx = tid; idx < KB_NB * KB_NB; idx += nt) Tm[idx] = czero();
    for (int r = R0 + tid; r < N; r += nt) xc[r] = W_(r, p0);
    if (!tm.sync(ctx)) return false;       // (nobody stores the finished column over the raw one before every workgroup has read it)
    for (int j = 0; j < KB_NB; ++j) {
        const int k = p0 + j;
        if (j > 0) {
            // ---- w1 = V_j^H x, w2 = T_j^H w1, x -= V_j w2 (replicated, x in LDS)
            for (int t = ctx.wave(); t < j; t += nw) {
                cd acc = team_dotc4(ctx, p0 + t + 1, N, [&](int r) { return hess_vt(W, ld, p0, r, t); }, xc);
                acc = ctx.wave_sum(acc);
                if (lane == 0) w1[t] = acc;
            }
            ctx.sync_lds();
            for (int t = tid; t < j; t += nt) {
                cd acc = czero();
                for (int s2 = 0; s2 <= t; ++s2) cfmac(acc, T_(s2, t), w1[s2]);
                w2[t] = acc;
            }
            ctx.sync_lds();
            team_sweep8(ctx, N - p0 - 1,
                [&](int q, int g, cd& acc, cd& unused) {
                    (void)unused;
                    const int r = p0 + 1 + q;
                    int t = g;
                    for (; t + KB_SWEEP_G < j; t += 2 * KB_SWEEP_G) {
                        const cd v0 = hess_vt(W, ld, p0, r, t), v1 = hess_vt(W, ld, p0, r, t + KB_SWEEP_G);
                        acc = acc + v0 * w2[t];
                        acc = acc + v1 * w2[t + KB_SWEEP_G];
                    }
                    if (t < j) acc = acc + hess_vt(W, ld, p0, r, t) * w2[t];
                },
                [&](int q, cd acc, cd unused) {
                    (void)unused;
                    const int r = p0 + 1 + q;
                    xc[r] = xc[r] - acc;
                });
            ctx.sync_lds();
        }
        // ---- reflector from rows k+1..N-1; the finished column to W (every workgroup, the same values)
        double beta;
        cd tau;
        const bool nextcol = j + 1 < KB_NB;
        cd pra = czero();                                          // row k+1 of V (columns t < j), fetched under the generator
        if (C::WS > 1 && tid < j) pra = hess_vt(W, ld, p0, k + 1, tid);
        // w1 = V_j^H v (replicated): v = [1; s x], the dots with the raw tail in front of the generator's barrier (larfg_front)
        const cd sh = larfg_front(ctx, N - k - 1, xc + k + 1, beta, tau, [&]() {
            for (int t = ctx.wave(); t < j; t += nw) {
                cd acc = team_dotc4(ctx, k + 2, N, [&](int r) { return hess_vt(W, ld, p0, r, t); }, xc);
                acc = ctx.wave_sum(acc);
                if (lane == 0) w1[t] = acc;
            }
        });
        for (int r = R0 + tid; r <= k; r += nt) W_(r, k) = xc[r];
        for (int r = k + 2 + tid; r < N; r += nt) W_(r, k) = xc[r];                 // (the entries this thread has just scaled)
        if (tid == 0) {
            W_(k + 1, k) = mk(beta, 0.0);
            xc[k + 1] = mk(1.0, 0.0);
            if (writer) tauh[j] = tau;
        }
        if (C::WS > 1) { if (tid < j) { ra[tid] = pra; w1[tid] = conj(pra) + w1[tid] * sh; } }
        else for (int t = tid; t < j; t += nt) { ra[t] = hess_vt(W, ld, p0, k + 1, t); w1[t] = conj(ra[t]) + w1[t] * sh; }
        ctx.sync_lds();
        // ---- the corrections Y_j w1 of y and of the next column, for the owned rows (Y(:, j-1) from LDS)
        team_sweep8(ctx, nownc * C::WS,
            [&](int q, int g, cd& hp, cd& np) {
                const int r = R0 + (role + (q / C::WS) * T) * C::WS + (q % C::WS);
                if (r >= N) return;
                int t = g;
                for (; t + KB_SWEEP_G < j; t += 2 * KB_SWEEP_G) {
                    const int t2 = t + KB_SWEEP_G;
                    const cd yrt = Y_(r, t), yr2 = (t2 == j - 1) ? yl[r] : Y_(r, t2);
                    hp = hp + yrt * w1[t];
                    if (nextcol) np = np + yrt * conj(ra[t]);
                    hp = hp + yr2 * w1[t2];
                    if (nextcol) np = np + yr2 * conj(ra[t2]);
                }
                if (t < j) {
                    const cd yrt = (t == j - 1) ? yl[r] : Y_(r, t);
                    hp = hp + yrt * w1[t];
                    if (nextcol) np = np + yrt * conj(ra[t]);
                }
            },
            [&](int q, cd hp, cd np) {
                const int r = R0 + (role + (q / C::WS) * T) * C::WS + (q % C::WS);
                if (r >= N) return;
                sp[r] = hp;
                sp[N + r] = np;
            });
        // ---- y = tau (A0[p0+1:, k+1:] v - Y_j w1): the one pass over A0 of this column, owned chunks of rows
        const int ncols = N - k - 1;
        const int xo = (j & 1) * 2 * N;       // two exchange regions in turn: a fast workgroup is one column ahead of the slowest reader at most
        for (int k0 = 0; k0 < nownc; k0 += zrc) {
            const int k1 = (k0 + zrc < nownc) ? k0 + zrc : nownc;
            team_row_product(ctx, tm, NR, ncols, &W_(R0, k + 1), ld, xc + k + 1, zp, zr, k0, k1);
            ctx.sync_lds();
            for (int q = tid; q < (k1 - k0) * C::WS; q += nt) {
                const int r = R0 + (role + (k0 + q / C::WS) * T) * C::WS + (q % C::WS);
                if (r >= N) continue;
                const cd h = team_row_sum(zp, zr, q);
                const cd y = tau * (h - sp[r]);
                const cd nx = nextcol ? (W_(r, k + 1) - sp[N + r]) - y : czero();   // ... - Y(r, j) conj(V(k+1, j)), V(k+1, j) = 1
                if (T == 1) {
                    Y_(r, j) = y;
                    yn[r] = y;                                 // (y_{j-1} in yl is still being read by the sweep's partners: a second array)
                    if (nextcol) xn[r] = nx;
                } else {
                    tm.put(xo + r, y);
                    if (nextcol) tm.put(xo + N + r, nx);
                }
            }
            ctx.sync_lds();
        }
        // ---- T(0:j, j) = -tau T_j w1, T(j, j) = tau
        for (int t = tid; t <= j; t += nt) {
            if (t == j) T_(j, j) = tau;
            else {
                cd acc = czero();
                for (int s2 = t; s2 < j; ++s2) cfma(acc, T_(t, s2), w1[s2]);
                T_(t, j) = -(tau * acc);
            }
        }
        if (!tm.sync(ctx)) return false;
        if (T > 1)
            for (int r = R0 + tid; r < N; r += nt) {
                const cd y = tm.get(xo + r);
                Y_(r, j) = y;
                yl[r] = y;
                if (nextcol) xc[r] = tm.get(xo + N + r);
            }
        else
            for (int r = R0 + tid; r < N; r += nt) {
                yl[r] = yn[r];
                if (nextcol) xc[r] = xn[r];
            }
        ctx.sync_lds();
    }
    if (!writer) return true;
    ctx.sync();                              // (the stores of Y(:, NB-1) and of the last column of V are read below)
    // ---- VT = V T (all rows; rows <= p0 are zero) and MT = (Y^H V) T for the deferred left factor: one workgroup
    for (int idx = tid; idx < N * 2; idx += nt) {             // two threads per row: the even / the odd columns t
        const int r = idx >> 1, half = idx & 1;
        cd acc[KB_NB / 2];
#pragma unroll
        for (int q = 0; q < KB_NB / 2; ++q) acc[q] = czero();
        for (int s2 = 0; s2 < KB_NB; ++s2) {                  // T is upper triangular with explicit zeros below
            const cd v = hess_vt(W, ld, p0, r, s2);
#pragma unroll
            for (int q = 0; q < KB_NB / 2; ++q) cfma(acc[q], v, T_(s2, 2 * q + half));
        }
#pragma unroll
        for (int q = 0; q < KB_NB / 2; ++q) VT[r + (size_t)(2 * q + half) * ldvt] = acc[q];
    }
    ctx.sync();
    {
        // (Y^H V T)(u, t) = sum_r conj(Y(r,u)) VT(r,t): one thread per entry, the rows staged through LDS in slabs
        const int slab = (2 * N / KB_NB) < 1 ? 1 : (2 * N / KB_NB);
        cd* sy = sp;
        cd* sv = sp + (size_t)slab * KB_NB;
        const int u = tid % KB_NB, t = tid / KB_NB;
        cd acc = czero();
        for (int r0 = p0 + 1; r0 < N; r0 += slab) {
            const int rows = (N - r0 < slab) ? N - r0 : slab;
            for (int idx = tid; idx < rows * KB_NB; idx += nt) {
                const int rr = idx % rows, cc = idx / rows;
                sy[rr + cc * slab] = Y_(r0 + rr, cc);
                sv[rr + cc * slab] = VT[(r0 + rr) + (size_t)cc * ldvt];
            }
            ctx.sync();
            if (tid < KB_NB * KB_NB)
                for (int rr = 0; rr < rows; ++rr) cfmac(acc, sy[rr + u * slab], sv[rr + t * slab]);
            ctx.sync();
        }
        if (tid < KB_NB * KB_NB) MT[u + t * KB_NB] = acc;
        for (int e2 = tid + nt; e2 < KB_NB * KB_NB; e2 += nt) {      // (fewer threads than entries: the host simulation)
            const int uu = e2 % KB_NB, tt = e2 / KB_NB;
            cd a2 = czero();
            for (int r = p0 + 1; r < N; ++r) cfmac(a2, Y_(r, uu), VT[r + (size_t)tt * ldvt]);
            MT[uu + tt * KB_NB] = a2;
        }
    }
    ctx.sync();
    return true;
#undef W_
#undef Y_
#undef T_
}

}  // namespace kb
