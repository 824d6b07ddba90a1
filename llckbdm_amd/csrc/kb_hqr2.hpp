// Second-generation QR iteration on an upper Hessenberg matrix (eigenvalues only, active block only):
// part of the zgeev replacement, reference kbdm.py:192.
//
// What changed against kb_hqr_ms.hpp (round 1), and why:
//   * DOUBLE-SHIFT bulges.  Every bulge carries two shifts and is chased with 3-element Householder
//     reflectors (the zlaqr5 organisation; for complex matrices any two shifts may share a bulge).  Bulges
//     still follow each other three rows apart, so a chain of nb bulges applies 2 nb shifts per sweep: the
//     same shifts need half the chase intervals (the iteration is latency bound per interval, not per flop).
//   * REGISTER-SYSTOLIC strip replay.  The reflectors logged by a window step are replayed on the strips
//     right of / above the window by single wavefronts, 8 bulges x 8 columns (or rows) per wavefront: a lane
//     keeps the two rows its bulge currently straddles in registers, takes the next row from the lane of the
//     bulge ahead with one DPP shift (row p finished by bulge b at interval t is exactly the row bulge b+1
//     needs at interval t+1) and hands its finished row on the same way.  Only the leading bulge loads from
//     memory (prefetched four steps ahead) and only the trailing one stores: every strip element moves once
//     per window step, nothing is staged in LDS, no workgroup barrier is executed.
//   * LDS holds the window and the log only (no tile images): 73 KB at W = 56, so two workgroups fit a CU.
//   * The team (chase workgroup + helper workgroup on another CU) keeps round 1's protocol; the record
//     that crosses is the time-major reflector log.
#pragma once
#include "kb_hqr_ms.hpp"

namespace kb {

constexpr int KB2_NBMAX = 8;        // bulges in flight (2 shifts each)
constexpr int KB2_NSMAX = 16;       // shifts per sweep
constexpr int KB2_WIN_DEV = 56;     // LDS window of the device kernels (W = 64 measured 3 % faster on one member but is not validated)
constexpr int KB2_MS_MIN = 9;       // below this active size: single-shift sweeps (blocks of at most 8: every
                                    // element they touch is within 7 of the diagonal, i.e. inside the chase
                                    // workgroup's windows and its one top unit)
// Strip units: `unit` = (W + 1) / 8 (at most 8) columns of the right strip / rows of the top strip per wavefront;
// the eight unit tiles of a workgroup fill exactly the window image in LDS (W = 56: 8 x 7 x 57 elements).
// In a team the chase workgroup keeps the right units that the NEXT window needs and one top unit.
KB_HD int hqr2_unit(int W) { const int u = (W + 1) / 8; return u > 8 ? 8 : u; }

struct alignas(16) Refl3 {          // H = I - tau v v^H, v = (1, v2, v3); tau = 0: identity
    cd tau, v2, v3;
};

// zlarfg for a 3-vector (alpha; x1; x2), branch-light: no rescaling (the chase works on O(norm) data).
KB_HD void larfg3(cd& alpha, cd& x1, cd& x2, cd& tau) {
    const double xn2 = x1.x * x1.x + x1.y * x1.y + x2.x * x2.x + x2.y * x2.y;
    if (xn2 == 0.0 && alpha.y == 0.0) { tau = czero(); return; }
#if defined(__HIP_DEVICE_COMPILE__)
    const double t = fma(alpha.x, alpha.x, fma(alpha.y, alpha.y, xn2));
    double rs = __builtin_amdgcn_rsq(t);
    rs = rs * fma(-0.5 * t * rs, rs, 1.5);
    rs = rs * fma(-0.5 * t * rs, rs, 1.5);
    const double nrm = t * rs;
    const double beta = (alpha.x >= 0.0) ? -nrm : nrm;
    const double ib = (alpha.x >= 0.0) ? -rs : rs;
    tau = mk((beta - alpha.x) * ib, -alpha.y * ib);
    const double dr = alpha.x - beta, di = alpha.y;
    const double d2 = fma(dr, dr, di * di);
    double r = __builtin_amdgcn_rcp(d2);
    r = r * fma(-d2, r, 2.0);
    r = r * fma(-d2, r, 2.0);
    const double cr = dr * r, ci = di * r;            // conj(d) / |d|^2 = (cr, -ci)
    x1 = mk(x1.x * cr + x1.y * ci, x1.y * cr - x1.x * ci);
    x2 = mk(x2.x * cr + x2.y * ci, x2.y * cr - x2.x * ci);
    alpha = mk(beta, 0.0);
#else
    const double nrm = sqrt(alpha.x * alpha.x + alpha.y * alpha.y + xn2);
    const double beta = (alpha.x >= 0.0) ? -nrm : nrm;
    tau = mk((beta - alpha.x) / beta, -alpha.y / beta);
    const cd d = mk(alpha.x - beta, alpha.y);
    x1 = cdiv(x1, d);
    x2 = cdiv(x2, d);
    alpha = mk(beta, 0.0);
#endif
}

// One reflector on three values.  SIDE 0: from the left on a column (H^H x), SIDE 1: from the right on a
// row (y H).  The two differ by conjugating every parameter.
template <int SIDE>
KB_HD void apply3(const Refl3& rf, cd& x0, cd& x1, cd& x2) {
    const double sg = SIDE == 0 ? -1.0 : 1.0;
    const cd tc = mk(rf.tau.x, sg * rf.tau.y);
    const cd a2 = mk(rf.v2.x, sg * rf.v2.y), a3 = mk(rf.v3.x, sg * rf.v3.y);
    cd s = x0;
    cfma(s, a2, x1);
    cfma(s, a3, x2);
    s = tc * s;
    x0 = x0 - s;
    // x1 -= s * conj(a2), x2 -= s * conj(a3)
    x1.x = fma(-s.x, a2.x, x1.x); x1.x = fma(-s.y, a2.y, x1.x);
    x1.y = fma(s.x, a2.y, x1.y);  x1.y = fma(-s.y, a2.x, x1.y);
    x2.x = fma(-s.x, a3.x, x2.x); x2.x = fma(-s.y, a3.y, x2.x);
    x2.y = fma(s.x, a3.y, x2.y);  x2.y = fma(-s.y, a3.x, x2.y);
}

template <int SIDE>
KB_HD void apply3v(cd tau, cd v2, cd v3, cd& x0, cd& x1, cd& x2) {
    const double sg = SIDE == 0 ? -1.0 : 1.0;
    const cd tc = mk(tau.x, sg * tau.y);
    const cd a2 = mk(v2.x, sg * v2.y), a3 = mk(v3.x, sg * v3.y);
    cd s = x0;
    cfma(s, a2, x1);
    cfma(s, a3, x2);
    s = tc * s;
    x0 = x0 - s;
    x1.x = fma(-s.x, a2.x, x1.x); x1.x = fma(-s.y, a2.y, x1.x);
    x1.y = fma(s.x, a2.y, x1.y);  x1.y = fma(-s.y, a2.x, x1.y);
    x2.x = fma(-s.x, a3.x, x2.x); x2.x = fma(-s.y, a3.y, x2.x);
    x2.y = fma(s.x, a3.y, x2.y);  x2.y = fma(-s.y, a3.x, x2.y);
}

struct Win2Geom {
    int l, i, na, nb, t0, t1, nint, ws, we, wlen, bmin, bmax;
    int unit;       // lines per strip unit
    int nr_near;    // right units the chase workgroup keeps (team); the helper starts behind them
};

// LDS carve (after the block-reduction slots): shift block S (16 x 16), shifts, info words, window, log.
struct Hqr2Lds {
    cd* S;
    cd* sh;
    int* sinfo;
    cd* Hw;
    Refl3* logv;
    cd* dmy;        // zero triples for the lanes without an element (branch-free chase passes): KB2_DMY elements
    int* flag;
};
constexpr int KB2_DMY = 192 + 64 + 2 * 65;     // 3 per lane (row operations) + lane + {0, WP, 2 WP} (column operations), WP <= 65
KB_HD int hqr2_win_elems(int W) { return W * (W + 1); }
KB_HD int hqr2_log_entries(int W) { return W * KB2_NBMAX; }
KB_HD int hqr2_scratch_bytes(int W) {
    return (KB2_NSMAX * KB2_NSMAX + KB2_NSMAX) * (int)sizeof(cd) + 64 + hqr2_win_elems(W) * (int)sizeof(cd) +
           hqr2_log_entries(W) * (int)sizeof(Refl3) + KB2_DMY * (int)sizeof(cd) + 64;
}
template <class C>
KB_HD Hqr2Lds hqr2_lds(const C& ctx, int W) {
    Hqr2Lds L;
    L.S = reinterpret_cast<cd*>(ctx.scratch());
    L.sh = L.S + KB2_NSMAX * KB2_NSMAX;
    L.sinfo = reinterpret_cast<int*>(L.sh + KB2_NSMAX);
    L.flag = L.sinfo + 4;
    L.Hw = reinterpret_cast<cd*>(reinterpret_cast<char*>(L.sinfo) + 64);
    L.logv = reinterpret_cast<Refl3*>(L.Hw + hqr2_win_elems(W));
    L.dmy = reinterpret_cast<cd*>(L.logv + hqr2_log_entries(W));
    return L;
}

struct Team2Rec {                   // 64-byte record header, followed by the log (time-major, 8 bulges per interval)
    int l, i, nb, na, t0, t1, ws, we, bmin, bmax, nint, g;
    int unit, nr_near, pad[2];
};
KB_HD int team2_rec_bytes(int W) { return (int)sizeof(Team2Rec) + hqr2_log_entries(W) * (int)sizeof(Refl3); }

template <class C>
struct Team2 {
    TeamCtl* ctl;
    char* ring;
    int rec_bytes;
    unsigned g, g_batch;
    int failed;
    HSc1 A;
    int W;
    int ws_last[2] = {0x3fffffff, 0x3fffffff};     // first rows of the windows of the last two records
};

// ---- strip units.  Right unit u: columns we + 8 u .. + 7 (not beyond i), row operations on the window's rows.
//                    Top unit u: rows ws - 8 (u + 1) .. ws - 8 u - 1 (not above l), column operations on the
//                    window's columns.
KB_HD void strip_unit_counts(const Win2Geom& G, int& nru, int& ntu) {
    const int nright = (G.we <= G.i) ? G.i - G.we + 1 : 0;
    const int ntop = G.ws - G.l;
    nru = (nright + G.unit - 1) / G.unit;
    ntu = (ntop + G.unit - 1) / G.unit;
}

// Reference replay of one strip line (column q of the right strip / row q of the top strip): bulge-major
// (reflectors of different bulges act on disjoint rows whenever their time order is swapped, so they commute).
template <int SIDE, class ACC>
KB_HD void strip_line_ref(const ACC& A, const Win2Geom& G, const Refl3* logv, int q) {
    for (int b = G.bmin; b <= G.bmax; ++b)
        for (int tau = 0; tau < G.nint; ++tau) {
            const int p = G.l + G.t0 + tau - 3 * b;
            if (p < G.l || p > G.i - 1) continue;
            const Refl3 rf = logv[tau * KB2_NBMAX + (b - G.bmin)];
            const bool three = p + 2 <= G.i;
            cd x0 = SIDE == 0 ? A.get(p, q) : A.get(q, p);
            cd x1 = SIDE == 0 ? A.get(p + 1, q) : A.get(q, p + 1);
            cd x2 = three ? (SIDE == 0 ? A.get(p + 2, q) : A.get(q, p + 2)) : czero();
            apply3<SIDE>(rf, x0, x1, x2);
            if (SIDE == 0) { A.put(p, q, x0); A.put(p + 1, q, x1); if (three) A.put(p + 2, q, x2); }
            else { A.put(q, p, x0); A.put(q, p + 1, x1); if (three) A.put(q, p + 2, x2); }
        }
}

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ cd dpp_shr1(cd v) {
    return mk(DevCtx::dpp_f64<0x111>(v.x), DevCtx::dpp_f64<0x111>(v.y));      // row_shr:1, zero fill
}

// Buffer view of the matrix for the strip replay: a lane that has nothing to load / store passes an out-of-range
// offset (loads return 0, stores are dropped by the bounds check), so tile loads and stores are branch-free
// batches.  AUX = 0: ordinary accesses (one workgroup per member), 16: sc1 (team traffic, see HSc1).
template <int AUX>
struct HBuf {
    __amdgpu_buffer_rsrc_t rs;
    int ld;
    __device__ static HBuf make(cd* H, int ld_, int ncols) {
        HBuf a;
        a.rs = __builtin_amdgcn_make_buffer_rsrc(H, 0, (int)((size_t)ld_ * ncols * sizeof(cd)), 0x00020000);
        a.ld = ld_;
        return a;
    }
    __device__ kb_u4 get(int off) const { return __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUX); }
    __device__ void put(int off, kb_u4 q) const { __builtin_amdgcn_raw_buffer_store_b128(q, rs, off, 0, AUX); }
};
__device__ __forceinline__ HBuf<0> strip_view(const HPlain& A) { return HBuf<0>::make(A.H, A.ld, A.nc); }
__device__ __forceinline__ HBuf<16> strip_view(const HSc1& A) { HBuf<16> b; b.rs = A.rs; b.ld = A.ld; return b; }

__device__ __forceinline__ cd csel(bool c, cd a, cd b) { return mk(c ? a.x : b.x, c ? a.y : b.y); }

// One wavefront, one unit: lane = 8 * line + bulge (line = column of the right strip / row of the top strip).
//  1. the unit's tile (lines x window length) is fetched into this wavefront's LDS tile T in one batch of loads;
//  2. the reflectors are replayed as a register pipeline.  Step tau = interval t0 + tau of the window step: bulge b
//     sits at p = l + t0 + tau - 3 b and owns stream positions p, p+1 (registers x0, x1); p+2 arrives from the lane
//     of bulge b-1 (its finished position of the step before, one DPP shift) or, for the leading bulge, from the
//     tile; position p leaves finished: to the lane of bulge b+1, or back into the tile from the trailing bulge.  A
//     bulge that enters the block during the window step is carried two steps early with identity reflectors (its
//     log entries are zero), so rows l, l+1 reach it through the pipeline like every other row;
//  3. the tile is written back in one batch of stores.
// No workgroup barrier; loads and stores never interleave (a vmcnt wait on a load would wait for older stores).
template <int SIDE, int AUX>
__device__ __forceinline__ void strip_unit_dev(const HBuf<AUX>& A, const Win2Geom& G, const Refl3* __restrict__ logv,
                                               cd* __restrict__ T, int TP, int q0, int qhi, int lane) {
    const int nl = qhi - q0 + 1, wlen = G.wlen, ws = G.ws;
    const int OOB = 0x7FFFFFF0;
    // ---- 1. tile in (at most 8 elements per lane: 8 lines x 64 positions)
    kb_u4* T4 = reinterpret_cast<kb_u4*>(T);
    {
        kb_u4 v[8];
        int ta[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            // (line, position) of this lane's k-th element without div / mod: right strip - lanes along the positions
            // (contiguous rows of a column), k = line; top strip - 8 lanes along the lines (contiguous rows), 8 positions per k
            int ln, sp;
            if (SIDE == 0) { ln = k; sp = lane; }
            else { ln = lane & 7; sp = (lane >> 3) + 8 * k; }
            const bool ok = ln < nl && sp < wlen;
            const int e = SIDE == 0 ? (q0 + ln) * A.ld + ws + sp : (q0 + ln) + (ws + sp) * A.ld;
            ta[k] = ok ? ln * TP + sp : -1;
            v[k] = A.get(ok ? e * 16 : OOB);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (ta[k] >= 0) T4[ta[k]] = v[k];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // ---- 2. pipeline
    {
        const int bi = lane & 7, line = lane >> 3;
        const int nbw = G.bmax - G.bmin + 1;
        const bool mine = line < nl && bi < nbw;
        const bool has_succ = bi + 1 < nbw;
        const int pbase = G.l + G.t0 - 3 * (G.bmin + bi);
        const int l = G.l, i = G.i, nint = G.nint;
        const cd* Tl = T + line * TP - ws;               // Tl[s] = stream position s of this lane's line
        auto ldt = [&](int s, bool ok) -> cd {
            const bool v = ok && s >= l && s <= i;
            const cd x = Tl[v ? s : ws];
            return csel(v, x, czero());
        };
        cd* Tw = T + line * TP - ws;
        // STEADY window step: all eight bulges are in flight and stay inside the block for every interval of the step (the
        // interior of a sweep: most window steps of a large active block) - every position a lane touches is a valid tile
        // position, so the step needs no range tests, no zero fills and no guarded stores (a lane that does not store
        // writes the padding element of its line instead): 46 instead of 80 instructions per pipeline step.
        const bool steady = nbw == KB2_NBMAX && G.t0 >= 3 * (G.bmin + KB2_NBMAX - 1) &&
                            l + G.t0 - 3 * G.bmin + nint + 2 <= i;
        if (steady) {
            cd x0 = Tl[pbase], x1 = Tl[pbase + 1];
            cd nxt = Tl[pbase + 2];
            cd outp = czero();
            const cd* lg = reinterpret_cast<const cd*>(logv + bi);
            cd r_tau = lg[0], r_v2 = lg[1], r_v3 = lg[2];
            cd* const dump = T + (mine ? line : 0) * TP + (TP - 1);    // the padding element of this line (never a position:
                                                                        // wlen <= W = TP - 1); lanes without a line use line 0's
            const bool st_all = mine && !has_succ;             // the trailing bulge retires a position every step
            for (int tau = 0; tau < nint; ++tau) {
                const int p = pbase + tau;
                cd in2 = dpp_shr1(outp);
                in2 = csel(bi == 0 || tau == 0, nxt, in2);
                const cd c_tau = r_tau, c_v2 = r_v2, c_v3 = r_v3;
                {
                    const cd* e = lg + (tau + 1) * (KB2_NBMAX * 3);       // (one row beyond the step at the end: unused)
                    r_tau = e[0]; r_v2 = e[1]; r_v3 = e[2];
                }
                nxt = Tl[p + 3];                                           // (only the leader's is used; all are valid positions)
                apply3v<SIDE>(c_tau, c_v2, c_v3, x0, x1, in2);
                outp = x0; x0 = x1; x1 = in2;
                cd* dst = (st_all || (mine && tau + 1 == nint)) ? &Tw[p] : dump;
                *dst = outp;
            }
            const int pl = pbase + nint - 1;
            if (mine) { Tw[pl + 1] = x0; Tw[pl + 2] = x1; }
        } else {
            cd x0 = ldt(pbase, mine), x1 = ldt(pbase + 1, mine);
            cd nxt = ldt(pbase + 2, mine);                   // position p + 2 of step 0 (every bulge), later: leader only
            cd outp = czero();
            const cd* lg = reinterpret_cast<const cd*>(logv + bi);
            cd r_tau = lg[0], r_v2 = lg[1], r_v3 = lg[2];
            // Lanes outside their bulge's life carry zeros / stale finite values through identity reflectors (log entries
            // are zero there, positions beyond i read as zero), so the pipeline needs no activity selects: only the
            // stores are guarded.
            for (int tau = 0; tau < nint; ++tau) {
                const int p = pbase + tau;
                cd in2 = dpp_shr1(outp);
                in2 = csel(bi == 0 || tau == 0, nxt, in2);
                const cd c_tau = r_tau, c_v2 = r_v2, c_v3 = r_v3;
                {
                    const cd* e = lg + (tau + 1 < nint ? tau + 1 : tau) * (KB2_NBMAX * 3);   // next step's reflector and the
                    r_tau = e[0]; r_v2 = e[1]; r_v3 = e[2];                                  // leader's next position: LDS
                }                                                                            // latency off the chain
                nxt = ldt(p + 3, mine && bi == 0);
                apply3v<SIDE>(c_tau, c_v2, c_v3, x0, x1, in2);
                outp = x0; x0 = x1; x1 = in2;
                // position p is finished for this bulge: to the next bulge (one step later) or into the tile
                const bool pass = has_succ && tau + 1 < nint;
                if (mine && !pass && p >= l && p <= i) Tw[p] = outp;
            }
            // the two positions still in flight after the last interval of the window step
            const int pl = pbase + nint - 1;
            if (mine) {
                if (pl + 1 >= l && pl + 1 <= i) Tw[pl + 1] = x0;
                if (pl + 2 >= l && pl + 2 <= i) Tw[pl + 2] = x1;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // ---- 3. tile out
    {
        kb_u4 v[8];
        int go[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int ln, sp;
            if (SIDE == 0) { ln = k; sp = lane; }
            else { ln = lane & 7; sp = (lane >> 3) + 8 * k; }
            const bool ok = ln < nl && sp < wlen;
            const int e = SIDE == 0 ? (q0 + ln) * A.ld + ws + sp : (q0 + ln) + (ws + sp) * A.ld;
            go[k] = ok ? e * 16 : OOB;
            v[k] = T4[ok ? ln * TP + sp : 0];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) A.put(go[k], v[k]);
    }
}
#endif

// All strip units [ru_lo, ru_hi) right and [tu_lo, tu_hi) top of one window step.  Device: unit k of the list goes to
// wavefront k mod nwaves, which stages it in its own tile of the (idle) window image; no barrier unless sig_word is
// given (helper: the first round is reported through it).
template <class C, class ACC>
KB_HD void strips_run(const C& ctx, const ACC& A, const Win2Geom& G, const Refl3* logv, cd* Hw, int W, int ru_lo, int ru_hi,
                      int tu_lo, int tu_hi, unsigned* sig_word = nullptr, unsigned sig_value = 0) {
    const int nr = ru_hi > ru_lo ? ru_hi - ru_lo : 0, ntp = tu_hi > tu_lo ? tu_hi - tu_lo : 0;
    const int nunits = nr + ntp;
    const int unit = G.unit;
#if defined(__HIP_DEVICE_COMPILE__)
    const int nw = ctx.nwaves(), w = ctx.wave(), lane = ctx.lane();
    const auto B = strip_view(A);
    const int TP = W + 1;
    cd* T = Hw + w * unit * TP;
    bool sent = false;
    for (int k0 = 0; k0 < nunits || (sig_word && !sent); k0 += nw) {
        const int k = k0 + w;
        if (k < nunits && w < 8) {
            if (k < nr) {
                const int c0 = G.we + (ru_lo + k) * unit;
                int chi = c0 + unit - 1;
                if (chi > G.i) chi = G.i;
                strip_unit_dev<0>(B, G, logv, T, TP, c0, chi, lane);
            } else {
                const int u = tu_lo + (k - nr);
                int r0 = G.ws - (u + 1) * unit;
                const int rhi = G.ws - u * unit - 1;
                if (r0 < G.l) r0 = G.l;
                strip_unit_dev<1>(B, G, logv, T, TP, r0, rhi, lane);
            }
        }
        if (sig_word && !sent) { team_signal(ctx, sig_word, sig_value); sent = true; }
    }
#else
    (void)ctx; (void)Hw; (void)W;
    for (int k = 0; k < nunits; ++k) {
        if (k < nr) {
            const int c0 = G.we + (ru_lo + k) * unit;
            for (int q = c0; q < c0 + unit && q <= G.i; ++q) strip_line_ref<0>(A, G, logv, q);
        } else {
            const int u = tu_lo + (k - nr);
            int r0 = G.ws - (u + 1) * unit;
            const int rhi = G.ws - u * unit - 1;
            if (r0 < G.l) r0 = G.l;
            for (int q = r0; q <= rhi; ++q) strip_line_ref<1>(A, G, logv, q);
        }
    }
    if (sig_word) *sig_word = sig_value;
#endif
}

// The helper's share of one record: far right units (first round reported through near_done), then far top units.
template <class C>
KB_HD void team2_helper_record(const C& ctx, Team2<C>& tm, const Win2Geom& G, unsigned g, const Refl3* logv, cd* Hw) {
    int nru, ntu;
    strip_unit_counts(G, nru, ntu);
    strips_run(ctx, tm.A, G, logv, Hw, tm.W, G.nr_near, nru, 0, 0, &tm.ctl->near_done, g + 1);
    strips_run(ctx, tm.A, G, logv, Hw, tm.W, 0, 0, 1, ntu);
    team_signal(ctx, &tm.ctl->all_done, g + 1);
}

template <class C>
KB_HD void team2_helper_main(const C& ctx, Team2<C>& tm) {
#if defined(__HIP_DEVICE_COMPILE__)
    const Hqr2Lds L = hqr2_lds(ctx, tm.W);
    const int tid = ctx.tid(), nt = ctx.nthreads();
    for (unsigned g = 0;; ++g) {
        if (tid == 0) {
            int st = 0;                         // 1: record ready, 2: finished, 3: abort
            const unsigned long long t_start = wall_clock64();
            for (;;) {
                if (__hip_atomic_load(&tm.ctl->published, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > g) { st = 1; break; }
                if (__hip_atomic_load(&tm.ctl->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    st = (__hip_atomic_load(&tm.ctl->published, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > g) ? 1 : 2;
                    break;
                }
                if (__hip_atomic_load(&tm.ctl->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { st = 3; break; }
                if (wall_clock64() - t_start > 2000000000ull) {     // 20 s: the chaser is gone
                    __hip_atomic_store(&tm.ctl->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    st = 3;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            *L.flag = st;
        }
        ctx.sync();
        const int st = *L.flag;
        ctx.sync();
        if (st != 1) return;
        const char* rec = tm.ring + (size_t)(g % KB_TEAM_SLOTS) * tm.rec_bytes;
        __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(rec), 0, tm.rec_bytes, 0x00020000);
        Win2Geom G;
        {
            const kb_u4 h0 = __builtin_amdgcn_raw_buffer_load_b128(rr, 0, 0, 16);
            const kb_u4 h1 = __builtin_amdgcn_raw_buffer_load_b128(rr, 16, 0, 16);
            const kb_u4 h2 = __builtin_amdgcn_raw_buffer_load_b128(rr, 32, 0, 16);
            G.l = (int)h0.x; G.i = (int)h0.y; G.nb = (int)h0.z; G.na = (int)h0.w;
            G.t0 = (int)h1.x; G.t1 = (int)h1.y; G.ws = (int)h1.z; G.we = (int)h1.w;
            const kb_u4 h3 = __builtin_amdgcn_raw_buffer_load_b128(rr, 48, 0, 16);
            G.bmin = (int)h2.x; G.bmax = (int)h2.y; G.nint = (int)h2.z;
            G.unit = (int)h3.x; G.nr_near = (int)h3.y;
            G.wlen = G.we - G.ws;
        }
        const int nq = G.nint * KB2_NBMAX * 3;                       // 16-byte words of the log
        kb_u4* dst = reinterpret_cast<kb_u4*>(L.logv);
        for (int idx = tid; idx < nq; idx += nt)
            dst[idx] = __builtin_amdgcn_raw_buffer_load_b128(rr, (int)sizeof(Team2Rec) + idx * 16, 0, 16);
        ctx.sync();
        team2_helper_record(ctx, tm, G, g, L.logv, L.Hw);
    }
#else
    (void)ctx; (void)tm;
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
// The intervals [t0, t1) of one window step on the device.  The iteration is bound by instruction issue and LDS round
// trips (a lone wavefront issues one FP64 instruction per ~6 cycles whether dependent or not, two wavefronts on a SIMD
// share ~4.8 cycles per instruction, an LDS round trip is ~100 cycles: tools/ubench/fp64_issue.hip), so:
//   * every wavefront works on TWO bulges at once, one per half-wavefront (lanes 0..31 / 32..63): each instruction of a
//     pass, and the whole reflector generation, serves two bulges;
//   * wavefronts 0..3 ("near") own the 32 columns p..p+31 of their bulges' row operations and the 32 rows p-28..p+3
//     of their column operations; wavefronts 4..7 ("far", same bulge pair as wavefront w-4) take what lies beyond
//     (columns p+32.., rows ..p-29), which exists for the trailing bulges in phase 1 and the leading ones in phase 2:
//     one pass per wavefront and phase, LDS latencies of the two wavefronts of a SIMD overlap;
//   * TWO barriers per interval.  Phase 1: row operations.  Phase 2: column operations; the near lanes 29..31 then
//     hold rows p+1..p+3 of the new column p, drop them into a 3-element LDS slot, and every lane of the half forms
//     the NEXT reflector from it in registers, publishing it in the log (where the far wavefront picks it up after
//     the barrier).  A bulge that enters at row l reads its start vector from the window image in phase 2 of the
//     interval before;
//   * window addresses are lane-constant plus compile-time offsets (W is a template parameter).
// value of lane LN of the caller's half-wavefront (32 lanes), through the LDS crossbar (no LDS memory, no bank conflicts)
template <int LN>
__device__ __forceinline__ cd half_bcast(cd v) {
    constexpr int PAT = (LN << 5);                      // bit mode: and_mask 0, or_mask LN, xor_mask 0
    const int a = __builtin_amdgcn_ds_swizzle(__double2loint(v.x), PAT), b = __builtin_amdgcn_ds_swizzle(__double2hiint(v.x), PAT);
    const int c = __builtin_amdgcn_ds_swizzle(__double2loint(v.y), PAT), d = __builtin_amdgcn_ds_swizzle(__double2hiint(v.y), PAT);
    return mk(__hiloint2double(b, a), __hiloint2double(d, c));
}

template <int W, bool PROF>
__device__ __forceinline__ void chase2_intervals_dev(const DevCtx& ctx, cd* Hw, Refl3* logv, const cd* sh, cd* dmy, int l,
                                                     int i, int nb, int t0, int t1, int ws, int we, int bmin, MsStats* stats) {
    constexpr int WP = W + 1;
    // phase timers of wavefront 0 (PROF, KBDM_HQR_PROF=1 only): tload = phase 1, treplay = wait at barrier 1, tstore = phase 2
    // column operations, ntiles = reflector generation; the rest of cyc_chase is the wait at barrier 2
    const bool prof = PROF && stats != nullptr && ctx.tid() == 0;
    long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#define HW(i_, j_) Hw[((i_) - ws) + ((j_) - ws) * WP]
    const int wv = ctx.wave(), lane = ctx.lane();
    const int half = lane >> 5, hl = lane & 31;
    const bool near_w = wv < 4;
    // near wavefront w and far wavefront w + 4 share a SIMD: the far one serves the COMPLEMENTARY pair 3 - w, whose far
    // work falls into the phase in which the near wavefront of that SIMD has little to do
    const int bl = 2 * (near_w ? wv : 7 - wv) + half;    // bulge slot inside the window step (log column)
    const int b = bmin + bl;
    const int na = i - l + 1;
    const int cmax = (i < we - 1) ? i : we - 1;
    // The row / column operations are BRANCH-FREE: a lane without an element triple (and the third element of a triple
    // that ends at the block's last row / column) works on a zero triple of its own in `dmy` instead - a unitary
    // reflector leaves zeros zero - so the passes need no exec-mask bookkeeping around their loads and stores
    // (the loop is bound by instruction issue: ~550 wavefront instructions per SIMD and interval before this).
    //   dmy[3 lane .. 3 lane + 2]: the row-operation triple;  dmy[192 + lane + {0, WP, 2 WP}]: the column-operation triple
    cd* const dm1 = dmy + 3 * lane;
    cd* const dm2 = dmy + 192 + lane;
    auto active = [&](int t) { const int d = t - 3 * b; return wv < 8 && b <= nb - 1 && d >= 0 && d <= na - 2; };
    cd r_tau = czero(), r_v2 = czero(), r_v3 = czero();     // the reflector of the current interval (plain registers:
                                                            // a struct here ends up in scratch memory)
    double beta = 0.0;
    auto refl_from_window = [&](int t) {
        const int p = l + t - 3 * b;
        const bool three = p + 2 <= i;
        cd a0, a1, a2, tau;
        if (p == l) {
            const cd s1 = sh[2 * b], s2 = sh[2 * b + 1];
            const cd h11 = HW(l, l), h21 = HW(l + 1, l), h12 = HW(l, l + 1), h22 = HW(l + 1, l + 1);
            const cd h32 = three ? HW(l + 2, l + 1) : czero();
            const cd d2 = h11 - s2;
            const double s = cabs1(d2) + cabs1(h21);
            if (s == 0.0) { a0 = czero(); a1 = czero(); a2 = czero(); }
            else {
                const double is = 1.0 / s;
                const cd h21s = is * h21;
                a0 = h21s * h12 + (h11 - s1) * (is * d2);
                a1 = h21s * (h11 + h22 - s1 - s2);
                a2 = h21s * h32;
            }
        } else {
            a0 = HW(p, p - 1);
            a1 = HW(p + 1, p - 1);
            a2 = three ? HW(p + 2, p - 1) : czero();
        }
        larfg3(a0, a1, a2, tau);
        r_tau = tau; r_v2 = a1; r_v3 = a2;
        beta = a0.x;
    };
    // log entry (tt - t0, bl) <- rf, one 16-byte store from each of the lanes 0..2 of the half
    auto publish = [&](int tt) {
        if (hl < 3) {
            const cd v = csel(hl == 0, r_tau, csel(hl == 1, r_v2, r_v3));
            reinterpret_cast<cd*>(logv + (tt - t0) * KB2_NBMAX + bl)[hl] = v;
        }
    };
    if (near_w && active(t0)) { refl_from_window(t0); publish(t0); }
    ctx.sync();                                          // (reads of column p-1 before anybody's phase 1 rewrites it)
    for (int t = t0; t < t1; ++t) {
        const bool act = active(t);
        const int p = l + t - 3 * b;
        const bool three = p + 2 <= i;
        if (prof) c0 = KB_CLOCK();
        // ---- phase 1: rows p..p+2, columns p..min(i, we-1); column p-1 becomes (beta, 0, 0)
        if (near_w) {
            if (act && p > l && hl < 3 && (hl < 2 || three)) HW(p + hl, p - 1) = mk(hl == 0 ? beta : 0.0, 0.0);
            const bool on = act && p + hl <= cmax;
            cd* a = on ? &HW(p, p + hl) : dm1;
            cd* a2 = (on && three) ? a + 2 : dm1 + 2;
            cd x0 = a[0], x1 = a[1], x2 = *a2;
            apply3v<0>(r_tau, r_v2, r_v3, x0, x1, x2);
            a[0] = x0; a[1] = x1; *a2 = x2;
        } else if (__builtin_amdgcn_ballot_w64(act && p + 32 <= cmax) != 0) {
            { const cd* e = reinterpret_cast<const cd*>(logv + (t - t0) * KB2_NBMAX + bl); r_tau = e[0]; r_v2 = e[1]; r_v3 = e[2]; }
            const bool on = act && p + 32 + hl <= cmax;
            cd* a = on ? &HW(p, p + 32 + hl) : dm1;
            cd* a2 = (on && three) ? a + 2 : dm1 + 2;
            cd x0 = a[0], x1 = a[1], x2 = *a2;
            apply3v<0>(r_tau, r_v2, r_v3, x0, x1, x2);
            a[0] = x0; a[1] = x1; *a2 = x2;
        }
        if (prof) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); c1 = KB_CLOCK(); }
        ctx.sync();
        if (prof) c2 = KB_CLOCK();
        // ---- phase 2: columns p..p+2, rows ws..min(p+3, i)
        cd ynew = czero();                                // the lane's new column-p entry (near wavefronts)
        {
            const int rmax = (p + 3 < i) ? p + 3 : i;
            if (near_w) {
                const int r = p - 28 + hl;                // rows p-28 .. p+3; lanes 29..31 are rows p+1..p+3
                const bool on = act && r >= ws && r <= rmax;
                cd* a = on ? &HW(r, p) : dm2;
                cd* a2 = (on && three) ? a + 2 * WP : dm2 + 2 * WP;
                cd y0 = a[0], y1 = a[WP], y2 = *a2;
                apply3v<1>(r_tau, r_v2, r_v3, y0, y1, y2);
                a[0] = y0; a[WP] = y1; *a2 = y2;
                ynew = y0;                                // (zero on a lane without a row)
            } else if (__builtin_amdgcn_ballot_w64(act && p - 29 >= ws) != 0) {
                const int r = p - 60 + hl;                // rows p-60 .. p-29
                { const cd* e = reinterpret_cast<const cd*>(logv + (t - t0) * KB2_NBMAX + bl); r_tau = e[0]; r_v2 = e[1]; r_v3 = e[2]; }
                const bool on = act && r >= ws;
                cd* a = on ? &HW(r, p) : dm2;
                cd* a2 = (on && three) ? a + 2 * WP : dm2 + 2 * WP;
                cd y0 = a[0], y1 = a[WP], y2 = *a2;
                apply3v<1>(r_tau, r_v2, r_v3, y0, y1, y2);
                a[0] = y0; a[WP] = y1; *a2 = y2;
            }
        }
        if (prof) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); c3 = KB_CLOCK(); }
        // ---- the reflector of interval t + 1 (near wavefronts), published in the log
        if (near_w && t + 1 < t1) {
            const bool nxt = active(t + 1);
            // rows p+1..p+3 of the new column p sit in lanes 29..31 of the half: broadcast them (LDS crossbar)
            cd a0 = half_bcast<29>(ynew), a1 = half_bcast<30>(ynew), a2 = half_bcast<31>(ynew);
            if (nxt && act) {
                cd tau;
                if (p + 3 > i) a2 = czero();
                larfg3(a0, a1, a2, tau);
                r_tau = tau; r_v2 = a1; r_v3 = a2;
                beta = a0.x;
            } else if (nxt) {
                refl_from_window(t + 1);
            }
            if (nxt) publish(t + 1);
        }
        if (prof) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const long long c4 = KB_CLOCK();
            stats->cyc_tload += c1 - c0; stats->cyc_treplay += c2 - c1; stats->cyc_tstore += c3 - c2; stats->ntiles += c4 - c3;
        }
        ctx.sync();
    }
#undef HW
}
#endif

#if !defined(__HIP_DEVICE_COMPILE__)
// host simulation only (tests/hostsim): called after every window step with its geometry
inline void (*kb2_host_trace)(const Win2Geom& G, int team, int phase) = nullptr;
#endif

// ---- windowed chase of nb double-shift bulges (shifts sh[2b], sh[2b+1] drive bulge b).
template <class C, class ACC>
KB_HD void chase2_windowed(const C& ctx, const ACC& A, int l, int i, int nb, const cd* sh, int W, const Hqr2Lds& L,
                           MsStats* stats, Team2<C>* team = nullptr) {
#define HW(i_, j_) Hw[((i_) - ws) + ((j_) - ws) * WP]
    const int tid = ctx.tid(), nt = ctx.nthreads();
    const int na = i - l + 1;
    const int WP = W + 1;
    const int LW = (C::WS >= 64) ? 64 : W;                  // lanes per bulge in the row / column phases
    cd* Hw = L.Hw;
    Refl3* logv = L.logv;
    const int T = (na - 1) + 3 * (nb - 1);
    int t0 = 0;
    if (team) {
        Team2<C>& tm = *team;
        tm.g_batch = tm.g;
        // The helper may still be working on the last two steps of the sweep before (everything older is complete: the
        // chase workgroup waited for near_done at its last step).  Their far right units and top units lie in the columns
        // from their window's first row on: when that reaches into the first window of THIS sweep (small active blocks;
        // the exceptional-shift path gets here within a few microseconds), the window must not be loaded before the
        // helper is through.
        const int wsmin = tm.ws_last[0] < tm.ws_last[1] ? tm.ws_last[0] : tm.ws_last[1];
        if (tm.g > 0 && !tm.failed && wsmin < l + W) {
            if (!team_wait(ctx, &tm.ctl->all_done, tm.g, tm.ctl, L.flag)) tm.failed = 1;
        }
        if (tm.failed) return;
    }
    while (t0 < T) {
        // ---- window of this step: first row = row above the topmost active bulge
        int bh0 = t0 / 3;
        if (bh0 > nb - 1) bh0 = nb - 1;
        const int kmin0 = l + t0 - 3 * bh0;
        int ws = kmin0 - 1;
        if (ws < l) ws = l;
        int we = ws + W;                                    // exclusive
        if (we > i + 1) we = i + 1;
        // ---- how many intervals fit: the lowest bulge must keep p+3 inside the window
        int t1 = t0;
        for (; t1 < T && t1 - t0 < W; ++t1) {
            int blo = 0;
            if (t1 - (na - 2) > 0) blo = (t1 - (na - 2) + 2) / 3;
            const int kmax = l + t1 - 3 * blo;
            const int reach = (kmax + 3 < i) ? kmax + 3 : i;
            if (reach > we - 1) break;
        }
        if (t1 == t0) t1 = t0 + 1;                          // unreachable for W >= 3 nb + 8
        const int nint = t1 - t0;
        int bmin = 0;
        if (t0 - (na - 2) > 0) bmin = (t0 - (na - 2) + 2) / 3;
        int bmax = (t1 - 1) / 3;
        if (bmax > nb - 1) bmax = nb - 1;
        // log, time-major; a zero entry is the identity (bulge not in flight / its virtual step at the bottom)
        {
            cd* lz = reinterpret_cast<cd*>(logv);
            for (int idx = tid; idx < nint * KB2_NBMAX * 3; idx += nt) lz[idx] = czero();
        }
        // ---- (a) load the diagonal window
        const long long c_a = KB_CLOCK();
        const int wlen = we - ws;
        if (C::WS >= 64) {                                  // lane = row, wavefronts stride the columns: no div / mod per element
            const int r = tid & 63;
            if (r < wlen)
                for (int c = tid >> 6; c < wlen; c += nt >> 6) Hw[r + c * WP] = A.get(ws + r, ws + c);
        } else {
            for (int idx = tid; idx < wlen * wlen; idx += nt) {
                const int r = idx % wlen, c = idx / wlen;
                Hw[r + c * WP] = A.get(ws + r, ws + c);
            }
        }
        ctx.sync();
        const long long c_b = KB_CLOCK();
        // ---- (b) chase inside the window, logging the reflectors
#if defined(__HIP_DEVICE_COMPILE__)
        // the device chase is compiled for W = KB2_WIN_DEV (window addresses are lane-constant plus compile-time offsets);
        // the host library passes exactly this window (the generic code around it, and the host simulation, take any W)
        if (stats) chase2_intervals_dev<KB2_WIN_DEV, true>(ctx, Hw, logv, sh, L.dmy, l, i, nb, t0, t1, ws, we, bmin, stats);
        else chase2_intervals_dev<KB2_WIN_DEV, false>(ctx, Hw, logv, sh, L.dmy, l, i, nb, t0, t1, ws, we, bmin, nullptr);
#else
        const int cmax = (i < we - 1) ? i : we - 1;
        for (int t = t0; t < t1; ++t) {
            int b_hi = t / 3;
            if (b_hi > nb - 1) b_hi = nb - 1;
            int b_lo = 0;
            if (t - (na - 2) > 0) b_lo = (t - (na - 2) + 2) / 3;
            for (int b = b_lo + tid; b <= b_hi; b += nt) {
                const int p = l + t - 3 * b;
                const bool three = p + 2 <= i;
                cd a0, a1, a2, tau;
                if (p == l) {
                    // first column of (H - s1)(H - s2), scaled (zlaqr1)
                    const cd s1 = sh[2 * b], s2 = sh[2 * b + 1];
                    const cd h11 = HW(l, l), h21 = HW(l + 1, l), h12 = HW(l, l + 1), h22 = HW(l + 1, l + 1);
                    const cd h32 = three ? HW(l + 2, l + 1) : czero();
                    const cd d2 = h11 - s2;
                    const double s = cabs1(d2) + cabs1(h21);
                    if (s == 0.0) { a0 = czero(); a1 = czero(); a2 = czero(); }
                    else {
                        const double is = 1.0 / s;
                        const cd h21s = is * h21;
                        a0 = h21s * h12 + (h11 - s1) * (is * d2);
                        a1 = h21s * (h11 + h22 - s1 - s2);
                        a2 = h21s * h32;
                    }
                } else {
                    a0 = HW(p, p - 1);
                    a1 = HW(p + 1, p - 1);
                    a2 = three ? HW(p + 2, p - 1) : czero();
                }
                larfg3(a0, a1, a2, tau);
                if (p > l) {
                    HW(p, p - 1) = a0;
                    HW(p + 1, p - 1) = czero();
                    if (three) HW(p + 2, p - 1) = czero();
                }
                Refl3 rf;
                rf.tau = tau; rf.v2 = a1; rf.v3 = a2;
                logv[(t - t0) * KB2_NBMAX + (b - bmin)] = rf;
            }
            ctx.sync();
            const int nbk = b_hi - b_lo + 1;
            const Refl3* lt = logv + (t - t0) * KB2_NBMAX - bmin;
            // rows p..p+2 ; columns p..min(i, we-1)
            for (int idx = tid; idx < nbk * LW; idx += nt) {
                const int b = b_lo + idx / LW, o = idx % LW;
                const int p = l + t - 3 * b;
                const int j = p + o;
                if (j <= cmax) {
                    const Refl3 rf = lt[b];
                    const bool three = p + 2 <= i;
                    cd x0 = HW(p, j), x1 = HW(p + 1, j), x2 = three ? HW(p + 2, j) : czero();
                    apply3<0>(rf, x0, x1, x2);
                    HW(p, j) = x0;
                    HW(p + 1, j) = x1;
                    if (three) HW(p + 2, j) = x2;
                }
            }
            ctx.sync();
            // columns p..p+2 ; rows ws..min(p+3, i)
            for (int idx = tid; idx < nbk * LW; idx += nt) {
                const int b = b_lo + idx / LW, o = idx % LW;
                const int p = l + t - 3 * b;
                const int rmax = (p + 3 < i) ? p + 3 : i;
                const int r = ws + o;
                if (r <= rmax) {
                    const Refl3 rf = lt[b];
                    const bool three = p + 2 <= i;
                    cd y0 = HW(r, p), y1 = HW(r, p + 1), y2 = three ? HW(r, p + 2) : czero();
                    apply3<1>(rf, y0, y1, y2);
                    HW(r, p) = y0;
                    HW(r, p + 1) = y1;
                    if (three) HW(r, p + 2) = y2;
                }
            }
            ctx.sync();
        }
#endif
        const long long c_c = KB_CLOCK();
        Win2Geom G;
        G.l = l; G.i = i; G.na = na; G.nb = nb; G.t0 = t0; G.t1 = t1; G.nint = nint; G.ws = ws; G.we = we;
        G.wlen = wlen; G.bmin = bmin; G.bmax = bmax;
        G.unit = hqr2_unit(W);
        {
            // right units the chase workgroup keeps in a team: the columns that enter the next window of this sweep
            int adv = 0;
            if (t1 < T) {
                int bh1 = t1 / 3;
                if (bh1 > nb - 1) bh1 = nb - 1;
                int ws1 = l + t1 - 3 * bh1 - 1;
                if (ws1 < l) ws1 = l;
                int we1 = ws1 + W;
                if (we1 > i + 1) we1 = i + 1;
                adv = we1 - we;
            }
            G.nr_near = (adv + G.unit - 1) / G.unit;
        }
        if (team) {
            // ---- publish the record (geometry + log) first: the helper's far units do not depend on the window image
            Team2<C>& tm = *team;
            char* rec = tm.ring + (size_t)(tm.g % KB_TEAM_SLOTS) * tm.rec_bytes;
#if defined(__HIP_DEVICE_COMPILE__)
            __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(rec, 0, tm.rec_bytes, 0x00020000);
            if (tid < 4) {
                kb_u4 h;
                if (tid == 0) h = kb_u4{(unsigned)l, (unsigned)i, (unsigned)nb, (unsigned)na};
                else if (tid == 1) h = kb_u4{(unsigned)t0, (unsigned)t1, (unsigned)ws, (unsigned)we};
                else if (tid == 2) h = kb_u4{(unsigned)bmin, (unsigned)bmax, (unsigned)nint, tm.g};
                else h = kb_u4{(unsigned)G.unit, (unsigned)G.nr_near, 0u, 0u};
                __builtin_amdgcn_raw_buffer_store_b128(h, rr, tid * 16, 0, 16);
            }
            const int nq = nint * KB2_NBMAX * 3;
            const kb_u4* src = reinterpret_cast<const kb_u4*>(logv);
            for (int idx = tid; idx < nq; idx += nt)
                __builtin_amdgcn_raw_buffer_store_b128(src[idx], rr, (int)sizeof(Team2Rec) + idx * 16, 0, 16);
#else
            (void)rec;
#endif
            team_signal(ctx, &tm.ctl->published, tm.g + 1);
            tm.ws_last[0] = tm.ws_last[1];
            tm.ws_last[1] = ws;
        }
        // ---- (c) store the window back
        if (C::WS >= 64) {
            const int r = tid & 63;
            if (r < wlen)
                for (int c = tid >> 6; c < wlen; c += nt >> 6) A.put(ws + r, ws + c, Hw[r + c * WP]);
        } else {
            for (int idx = tid; idx < wlen * wlen; idx += nt) {
                const int r = idx % wlen, c = idx / wlen;
                A.put(ws + r, ws + c, Hw[r + c * WP]);
            }
        }
        ctx.sync();
        const long long c_d = KB_CLOCK();
        int nru, ntu;
        strip_unit_counts(G, nru, ntu);
        if (!team) {
            strips_run(ctx, A, G, logv, Hw, W, 0, nru, 0, ntu);
        } else {
            Team2<C>& tm = *team;
#if !defined(__HIP_DEVICE_COMPILE__)
            team2_helper_record(ctx, tm, G, tm.g, logv, Hw);    // host simulation: the helper's share runs inline
#endif
            // the near units of this step touch elements the helper wrote at step g - 1 (its first round of far
            // right units) and earlier; at the first step of a batch everything before (top strips of the
            // previous batch reach down to this window's columns)
            bool ok = true;
            if (tm.g > 0 && !tm.failed) {
                // (a step that keeps more than 6 right units reaches beyond the helper's first round of the step before)
                if (tm.g == tm.g_batch || G.nr_near > 6) ok = team_wait(ctx, &tm.ctl->all_done, tm.g, tm.ctl, L.flag);
                else ok = team_wait(ctx, &tm.ctl->near_done, tm.g, tm.ctl, L.flag);
            }
            if (!ok) tm.failed = 1;
            if (!tm.failed)
                strips_run(ctx, A, G, logv, Hw, W, 0, nru < G.nr_near ? nru : G.nr_near, 0, ntu < 1 ? ntu : 1);
            tm.g++;
        }
        ctx.sync();
#if !defined(__HIP_DEVICE_COMPILE__)
        if (kb2_host_trace) kb2_host_trace(G, team != nullptr, 0);
#endif
        if (stats && tid == 0) {
            const long long c_e = KB_CLOCK();
            stats->small_steps++;
            stats->cyc_load += c_b - c_a; stats->cyc_chase += c_c - c_b;
            stats->cyc_store += c_d - c_c; stats->cyc_strip += c_e - c_d;
        }
        t0 = t1;
    }
#undef HW
}

#if defined(__HIP_DEVICE_COMPILE__)
template <int K>
__device__ __forceinline__ cd quad_bcast(cd v) {
    constexpr int CTRL = K | (K << 2) | (K << 4) | (K << 6);            // quad_perm [K,K,K,K]
    return mk(DevCtx::dpp_f64<CTRL>(v.x), DevCtx::dpp_f64<CTRL>(v.y));
}
__device__ __forceinline__ double quad_sum(double v) {
    v += DevCtx::dpp_f64<0xB1>(v);      // quad_perm [1,0,3,2]
    v += DevCtx::dpp_f64<0x4E>(v);      // quad_perm [2,3,0,1]
    return v;
}

// Ehrlich-Aberth on Hyman's recurrence for exactly 16 roots (the shifts of a full sweep), FOUR LANES PER ROOT:
// lane 4 r + s keeps the partial column sums  sum_{i<=j} u_i T(i,j)  of the columns j = s mod 4 (row-oriented
// accumulation: when u_i is known, every later column takes its term, so the dependent chain of one level is one
// complex FMA + the division by the subdiagonal instead of a length-j dot product); the finished column is
// broadcast inside the quad by DPP.  16 accumulators per root instead of 64 registers of recurrence vectors.
// Same stopping rule, start values and failure semantics as aberth_eigs.  T: LDS, column-major, ld = 16.
__device__ __forceinline__ bool aberth16_quad(const DevCtx& c, const cd* __restrict__ T, cd* z, cd* zw, int maxit,
                                              int* iters) {
    constexpr int n = 16;
    const int lane = c.lane(), r = lane >> 2, sub = lane & 3;
    cd* rinv = zw + n;
    double sc = 0.0;
    int bad = 0;
    for (int idx = lane; idx < n * n; idx += 64) {
        const int rr = idx % n, cc = idx / n;
        if (rr <= cc + 1) sc = fmax(sc, cabs1(T[rr + cc * n]));
    }
    sc = c.wave_max(sc);
    if (lane < n) {
        z[lane] = T[lane + lane * n];
        if (lane < n - 1) {
            const cd h = T[lane + 1 + lane * n];
            if (is_zero(h)) bad = 1;
            else rinv[lane] = cdiv(mk(1.0, 0.0), h);
        }
    }
    c.lds_fence();
    if (lane < n) {
        cd zr = z[lane];
        for (int k = 0; k < lane; ++k)
            if (cabs1(zr - z[k]) <= 1e-8 * sc) {
                const double a = 1e-4 * sc * (double)(lane + 1);
                zr = zr + mk(a * (1.0 - 0.125 * k), a * 0.0625 * (k + 1));
            }
        zw[lane] = zr;
    }
    c.lds_fence();
    if (lane < n) z[lane] = zw[lane];
    c.lds_fence();
    if (c.wave_max(bad) != 0 || !(sc > 0.0)) return false;
    bool frozen = false, conv_all = false;
    int it = 0;
    for (; it < maxit && !conv_all; ++it) {
        const cd zr = z[r];
        cd su0 = czero(), su1 = czero(), su2 = czero(), su3 = czero();
        cd sd0 = czero(), sd1 = czero(), sd2 = czero(), sd3 = czero();
        cd u = mk(1.0, 0.0), d = czero(), acc = czero(), dacc = czero();
#define KB2_AB_ROW(J, Q, SU, SD)                                                     \
        if (4 * (Q) + 3 >= (J)) {                                                    \
            const int cc = 4 * (Q) + sub;                                            \
            const cd t = (cc >= (J)) ? T[(J) + cc * n] : czero();                    \
            cfma(SU, u, t);                                                          \
            cfma(SD, d, t);                                                          \
        }
#define KB2_AB_LEVEL(J)                                                              \
        {                                                                            \
            KB2_AB_ROW(J, 0, su0, sd0) KB2_AB_ROW(J, 1, su1, sd1) KB2_AB_ROW(J, 2, su2, sd2) KB2_AB_ROW(J, 3, su3, sd3) \
            cd a = ((J) >> 2) == 0 ? su0 : ((J) >> 2) == 1 ? su1 : ((J) >> 2) == 2 ? su2 : su3;   \
            cd b = ((J) >> 2) == 0 ? sd0 : ((J) >> 2) == 1 ? sd1 : ((J) >> 2) == 2 ? sd2 : sd3;   \
            cfma(a, -zr, u);                                                         \
            cfma(b, -zr, d);                                                         \
            b = b - u;                                                               \
            acc = quad_bcast<(J) & 3>(a);                                            \
            dacc = quad_bcast<(J) & 3>(b);                                           \
            if ((J) < n - 1) {                                                       \
                const cd ri = rinv[(J)];                                             \
                u = -(acc * ri);                                                     \
                d = -(dacc * ri);                                                    \
            }                                                                        \
        }
        KB2_AB_LEVEL(0) KB2_AB_LEVEL(1) KB2_AB_LEVEL(2) KB2_AB_LEVEL(3) KB2_AB_LEVEL(4) KB2_AB_LEVEL(5) KB2_AB_LEVEL(6) KB2_AB_LEVEL(7)
        KB2_AB_LEVEL(8) KB2_AB_LEVEL(9) KB2_AB_LEVEL(10) KB2_AB_LEVEL(11) KB2_AB_LEVEL(12) KB2_AB_LEVEL(13) KB2_AB_LEVEL(14) KB2_AB_LEVEL(15)
#undef KB2_AB_LEVEL
#undef KB2_AB_ROW
        cd dz = czero();
        int open_ = 0;
        // Aberth sum over the other roots, four per lane
        cd sum = czero();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 4 * q + sub;
            const cd df = zr - z[k];
            const double qq = 1.0 / abs2(df);
            if (k != r) sum = sum + mk(df.x * qq, -df.y * qq);
        }
        sum = mk(quad_sum(sum.x), quad_sum(sum.y));
        if (!frozen) {
            if (!is_zero(acc)) {
                const cd nw = cdiv(acc, dacc);
                dz = cdiv(nw, mk(1.0, 0.0) - nw * sum);
            }
            if (!(cabs1(dz) < 1e300)) { bad = 1; dz = czero(); }
            if (cabs1(dz) <= 4.0 * KB_ULP * fmax(cabs1(zr), 0.015625 * sc)) frozen = true;
            else open_ = 1;
        }
        c.lds_fence();                       // every lane has read the old roots
        if (sub == 0) z[r] = zr - dz;
        c.lds_fence();
        conv_all = c.wave_max(open_) == 0;
        if (c.wave_max(bad) != 0) return false;
    }
    if (iters) *iters = it;
    return conv_all;
}
#endif

// The shift solvers are cold (once per sweep) and register hungry: kept out of line on the device so that their
// register needs do not inflate the allocation of the chase / replay loops (two workgroups per CU need <= 128 VGPRs).
#if defined(__HIP_DEVICE_COMPILE__)
#define KB2_COLD __attribute__((noinline))
#else
#define KB2_COLD
#endif
template <class C>
KB2_COLD KB_HD void hqr2_shifts(const C& ctx, int ns, cd* S, cd* sh, cd* aws, int* sinfo, MsStats* stats) {
    WaveCtx<C> wc{ctx, nullptr, 0};
    bool ok = false;
    if (ns >= 3) {
        int iters = 0;
#if defined(__HIP_DEVICE_COMPILE__)
        if (ns == 16) ok = aberth16_quad(ctx, S, sh, aws, 60, &iters);
#else
        if (ns == 16) ok = aberth_eigs(wc, ns, S, ns, sh, aws, aws + ns * ns, aws + 2 * ns * ns, 60);
#endif
        else if (ns == 8) ok = aberth_eigs_reg<8>(wc, S, ns, sh, aws, 40, &iters);
        else ok = aberth_eigs(wc, ns, S, ns, sh, aws, aws + ns * ns, aws + 2 * ns * ns, 60);
        if (stats && ctx.tid() == 0) { stats->ab_calls++; stats->ab_iters += iters; if (!ok) stats->ab_fail++; }
    }
    if (!ok) hqr_eigvals(wc, ns, S, ns, sh, sinfo);
}
template <class C>
KB2_COLD KB_HD void hqr2_single_sweep(const C& ctx, cd* H, int ld, int l, int i, int kdefl) {
    single_shift_sweep(ctx, H, ld, l, i, kdefl);
}

// ---- driver: deflation scan, shifts, sweeps (the structure of hqr_eigvals_ms; nb bulges = 2 nb shifts per sweep)
template <class C>
KB_HD void hqr2_eigvals(const C& ctx, int n, cd* H, int ld, cd* w, int* info, int nbmax, int win_w,
                        MsStats* stats = nullptr, Team2<C>* team = nullptr, int smode = 0) {
#define HH(i_, j_) H[(i_) + (size_t)(j_) * ld]
    const double ulp = KB_ULP;
    const double smlnum = KB_SAFMIN * ((double)n / ulp);
    const int tid = ctx.tid(), nt = ctx.nthreads();
    if (nbmax > KB2_NBMAX) nbmax = KB2_NBMAX;
    if (nbmax < 1) nbmax = 1;
    const Hqr2Lds L = hqr2_lds(ctx, win_w);
    cd* S = L.S;
    cd* sh = L.sh;
    int* sinfo = L.sinfo;
    cd* aws = L.Hw;                                   // Aberth workspace: the window image, idle between chases
    int fail = 0;
    bool bail = false;
    for (int idx = tid; idx < KB2_DMY; idx += nt) L.dmy[idx] = czero();   // the chase's zero triples (device passes)
    if (n == 1) {
        if (tid == 0) { w[0] = HH(0, 0); *info = 0; }
        ctx.sync();
        if (team) team_signal(ctx, &team->ctl->done, 1u);
        return;
    }
    ctx.sync();
    {   // a matrix that is not finite (an exactly singular U^{p-1}: Dsqi = inf) would only burn its 30 n sweeps: fail at once
        int nonfinite = 0;
        for (int k = tid; k < n; k += nt) {
            const cd a = HH(k, k), b = (k > 0) ? HH(k, k - 1) : czero();
            if (!((a.x - a.x == 0.0) && (a.y - a.y == 0.0) && (b.x - b.x == 0.0) && (b.y - b.y == 0.0))) nonfinite = 1;
        }
        nonfinite = ctx.block_max(nonfinite);
        if (nonfinite) {
            if (tid == 0) *info = 1;
            ctx.sync();
            if (team) team_signal(ctx, &team->ctl->done, 1u);
            return;
        }
    }
    const int itmax = 30 * (n > 10 ? n : 10);
    int kdefl = 0;
    int i = n - 1;
    const long long c_total0 = KB_CLOCK();
    while (i >= 0) {
        int l = 0;
        int done = 0;   // 1: H(i,i) converged, 2: 2x2 block solved
        for (int its = 0; its <= itmax; ++its) {
            const long long c_scan0 = KB_CLOCK();
            // ---- deflation scan: largest k in (l, i] with a negligible subdiagonal (zlahqr criterion)
            int kf = l;
            for (int k = l + 1 + tid; k <= i; k += nt) {
                const cd hkk1 = HH(k, k - 1);
                bool small_ = false;
                if (cabs1(hkk1) <= smlnum) small_ = true;
                else {
                    double tst = cabs1(HH(k - 1, k - 1)) + cabs1(HH(k, k));
                    if (tst == 0.0) {
                        if (k - 2 >= 0) tst += cabs1(HH(k - 1, k - 2));
                        if (k + 1 <= n - 1) tst += cabs1(HH(k + 1, k));
                    }
                    if (cabs1(hkk1) <= ulp * tst) {
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
            if (l >= i) { done = 1; break; }
            ctx.sync();
            if (stats && tid == 0) stats->cyc_scan += KB_CLOCK() - c_scan0;
            const int na = i - l + 1;
            if (na == 2) {
                if (tid == 0) {
                    cd z1, z2;
                    eig2x2(HH(l, l), HH(l, i), HH(i, l), HH(i, i), z1, z2);
                    w[l] = z1; w[i] = z2;
                }
                done = 2;
                break;
            }
            kdefl++;
            if (na < KB2_MS_MIN) {
                // A block of at most 8: all its eigenvalues at once, by one wavefront on a copy in LDS (the small single-shift
                // iteration that also backs the shift solver), instead of one global-memory sweep + scan per iteration of
                // this loop (~16 k cycles each, two or three per eigenvalue).  Eigenvalues only: H is not needed any more.
                const long long c0 = KB_CLOCK();
                for (int idx = tid; idx < na * na; idx += nt) {
                    const int r = idx % na, c = idx / na;
                    S[r + c * na] = (r <= c + 1) ? HH(l + r, l + c) : czero();
                }
                ctx.sync();
                if (ctx.wave() == 0) {
                    if (tid == 0) *sinfo = 0;
                    hqr2_shifts(ctx, na, S, sh, aws, sinfo, nullptr);      // Ehrlich-Aberth on Hyman's recurrence (all roots at
                }                                                          // once, lane-parallel), the small QR iteration behind it
                ctx.sync();
                for (int r = tid; r < na; r += nt) w[l + r] = sh[r];
                if (*sinfo != 0) fail = 1;
                if (stats && tid == 0) { stats->single_sweeps++; stats->cyc_single += KB_CLOCK() - c0; }
                done = 2;
                break;
            } else {
                const long long c_sh0 = KB_CLOCK();
                int nb = na / 6;                              // 2 nb shifts <= na / 3
                if (nb > nbmax) nb = nbmax;
                if (nb < 1) nb = 1;
                const int ns = (smode == 1) ? nb : 2 * nb;       // distinct shifts (smode 1: each one used twice)
                // ---- shifts
                if (kdefl % 6 == 0) {
                    // exceptional shifts (zlaqr0): h(ii,ii) + 0.75 |h(ii,ii-1)|, in pairs
                    for (int b = tid; b < ns; b += nt) {
                        const int ii = i - (b & ~1);
                        sh[b] = HH(ii, ii) + mk(0.75 * cabs1(HH(ii, ii - 1)), 0.0);
                    }
                } else {
                    const int r0 = i - ns + 1;
                    for (int idx = tid; idx < ns * ns; idx += nt) {
                        const int r = idx % ns, c = idx / ns;
                        S[r + c * ns] = (r <= c + 1) ? HH(r0 + r, r0 + c) : czero();
                    }
                    ctx.sync();
                    if (ctx.wave() == 0) hqr2_shifts(ctx, ns, S, sh, aws, sinfo, stats);
                }
                ctx.sync();
                if (smode == 1) {
                    cd keep = czero();
                    if (tid < ns) keep = sh[tid];
                    ctx.sync();
                    if (tid < ns) { sh[2 * tid] = keep; sh[2 * tid + 1] = keep; }
                    ctx.sync();
                }
                if (stats && tid == 0) stats->cyc_shift += KB_CLOCK() - c_sh0;
                const int T = (na - 1) + 3 * (nb - 1);
                if (team) {
                    chase2_windowed(ctx, team->A, l, i, nb, sh, win_w, L, stats, team);
#if defined(__HIP_DEVICE_COMPILE__)
                    // the scan / shift / small-block code reads the band with plain loads: drop whatever this
                    // CU's L1 still holds of it (the window was stored sc1)
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
                    if (team->failed) { fail = 1; bail = true; }
                } else {
                    chase2_windowed(ctx, HPlain{H, ld, n}, l, i, nb, sh, win_w, L, stats);
                }
                if (stats && tid == 0) { stats->intervals += T; stats->batches++; }
                if (bail) break;
            }
        }
        ctx.sync();
        if (bail) {                 // team protocol failure: report the diagonal, flag the member
            for (int r = tid; r <= i; r += nt) w[r] = HH(r, r);
            break;
        }
        if (done == 1) {
            if (tid == 0) w[i] = HH(i, i);
            i = l - 1;
        } else if (done == 2) {
            i = l - 1;
        } else {
            fail = 1;
            for (int r = l + tid; r <= i; r += nt) w[r] = HH(r, r);
            i = l - 1;
        }
        kdefl = 0;
        ctx.sync();
    }
    if (tid == 0) *info = fail;
    if (stats && tid == 0) stats->cyc_total += KB_CLOCK() - c_total0;
    ctx.sync();
    if (team) team_signal(ctx, &team->ctl->done, 1u);
#undef HH
}

}  // namespace kb
