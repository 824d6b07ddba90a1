// Multi-bulge (small-bulge multishift) complex QR iteration on an upper Hessenberg matrix,
// eigenvalues only, active block only (part of the zgeev replacement, reference kbdm.py:192).
//
// Why: the single-shift sweep of hqr_eigvals is a chain of ~1.5 n^2 dependent steps, each
// costing two workgroup barriers and two global-memory round trips -> latency bound
// (k_hqr was 45 % of the pipeline in the first profile).  Here ns shifts (eigenvalues of the
// trailing ns x ns block, computed by ONE wavefront in LDS) drive ns 2x2 bulges that are
// chased simultaneously, three rows apart (Braman/Byers/Mathias small-bulge chains; the
// spacing makes the reflectors commute, so the result equals ns consecutive single-shift
// sweeps with the same shifts).  One barrier interval now advances ns bulges: the dependent
// chain shrinks by ~ns, the work per barrier grows by ns.
// Small active blocks (< KB_MS_MIN) fall back to the single-shift sweep; 2x2 blocks are
// solved in closed form.
#pragma once
#include "kb_eig.hpp"

namespace kb {

constexpr int KB_MS_MIN = 10;     // below this active size: single-shift sweeps (blocks of at most 9:
                                  // every element they touch is within KB_TEAM_TOPB of the diagonal)
constexpr int KB_MS_NSMAX = 32;   // compile-time cap on simultaneous shifts
constexpr int KB_MS_BU = 4;       // bulges whose loads are batched together
constexpr int KB_MS_CU = 2;       // row/column chunks per thread batched together
constexpr int KB_MS_RG = 8;       // bulges replayed as independent chains in the strip update

// The lanes of ONE wavefront presented as a tiny workgroup (for the small shift solver).
template <class C>
struct WaveCtx {
    static constexpr int WS = C::WS;
    const C& c;
    char* scr;
    int scr_bytes;
    KB_HD int tid() const { return c.lane(); }
    KB_HD int nthreads() const { return C::WS; }
    KB_HD int lane() const { return c.lane(); }
    KB_HD int wave() const { return 0; }
    KB_HD int nwaves() const { return 1; }
    KB_HD void sync() const { c.wave_fence(); }
    KB_HD void wave_fence() const { c.wave_fence(); }
    KB_HD char* scratch() const { return scr; }
    KB_HD int scratch_bytes() const { return scr_bytes; }
    KB_HD double wave_sum(double v) const { return c.wave_sum(v); }
    KB_HD cd wave_sum(cd v) const { return c.wave_sum(v); }
    KB_HD double wave_max(double v) const { return c.wave_max(v); }
    KB_HD int wave_max(int v) const { return c.wave_max(v); }
    KB_HD double block_sum(double v) const { c.wave_fence(); return c.wave_sum(v); }
    KB_HD cd block_sum(cd v) const { c.wave_fence(); return c.wave_sum(v); }
    KB_HD double block_max(double v) const { c.wave_fence(); return c.wave_max(v); }
    KB_HD int block_max(int v) const { c.wave_fence(); return c.wave_max(v); }
};

template <int N>
struct KbInt {
    static constexpr int value = N;
};

struct MsRefl {
    cd t1;
    cd v2;
    cd t2;        // t1 * v2.  zlahqr keeps only its real part, which is valid while every
                  // subdiagonal is real; inside a multi-bulge batch H(i,i-1) is complex after
                  // the first bulge has left the bottom, so the general complex form is used.
    int k;        // row/column index of the bulge
    int pad;
};

struct MsStats {          // optional instrumentation (host simulation / KBDM_HQR_PROF=1)
    long long intervals, batches, single_sweeps, small_steps;
    long long cyc_scan, cyc_shift, cyc_load, cyc_chase, cyc_store, cyc_strip, cyc_single, cyc_total;
    long long cyc_tload, cyc_treplay, cyc_tstore, ntiles;
    long long ab_calls, ab_fail, ab_iters;   // Aberth shift solves / fallbacks to the QR solver / iterations (ns = 8)
};

#if defined(__HIP_DEVICE_COMPILE__)
#define KB_CLOCK() ((long long)clock64())
#else
#define KB_CLOCK() (0LL)
#endif

KB_HD int hqr_ms_scratch_bytes(int nsmax) {
    return (nsmax * nsmax + nsmax) * (int)sizeof(cd) + nsmax * (int)sizeof(MsRefl) + 64;
}

// Eigenvalues of the 2x2 block [[a,b],[c,d]].
KB_HD void eig2x2(cd a, cd b, cd c, cd d, cd& z1, cd& z2) {
    const cd s = 0.5 * (a + d);
    const cd p = 0.5 * (a - d);
    const cd disc = csqrt_(p * p + b * c);
    const cd e1 = s + disc, e2 = s - disc;
    const cd det = a * d - b * c;
    if (cabs1(e1) >= cabs1(e2)) {
        z1 = e1;
        z2 = is_zero(e1) ? e2 : cdiv(det, e1);
    } else {
        z1 = e2;
        z2 = cdiv(det, e2);
    }
}

// One single-shift QR sweep on the active block [l..i] (zlahqr body).  Same code path as
// hqr_eigvals; kept separate so that the multishift driver can use it for small blocks.
template <class C>
KB_HD void single_shift_sweep(const C& ctx, cd* H, int ld, int l, int i, int kdefl) {
#define HH(i_, j_) H[(i_) + (size_t)(j_) * ld]
    const double ulp = KB_ULP;
    const int tid = ctx.tid(), nt = ctx.nthreads();
    cd t;
    if (kdefl % 20 == 0) {
        t = mk(0.75 * cabs1(HH(i, i - 1)), 0.0) + HH(i, i);
    } else if (kdefl % 10 == 0) {
        t = mk(0.75 * cabs1(HH(l + 1, l)), 0.0) + HH(l, l);
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
    int mf = l;
    for (int mm = l + 1 + tid; mm <= i - 1; mm += nt) {
        const cd h11 = HH(mm, mm), h22 = HH(mm + 1, mm + 1);
        cd h11s = h11 - t;
        double h21 = cabs1(HH(mm + 1, mm));
        const double s = cabs1(h11s) + h21;
        h11s = mk(h11s.x / s, h11s.y / s);
        h21 = h21 / s;
        const double h10 = cabs1(HH(mm, mm - 1));
        if (h10 * h21 <= ulp * (cabs1(h11s) * (cabs1(h11) + cabs1(h22))))
            if (mm > mf) mf = mm;
    }
    mf = ctx.block_max(mf);
    const int ms = mf;
    cd v1, v2;
    {
        cd h11s = HH(ms, ms) - t;
        const cd h21 = HH(ms + 1, ms);
        const double s = cabs1(h11s) + cabs1(h21);
        v1 = mk(h11s.x / s, h11s.y / s);
        v2 = mk(h21.x / s, h21.y / s);
    }
    for (int k = ms; k <= i - 1; ++k) {
        if (k > ms) { v1 = HH(k, k - 1); v2 = HH(k + 1, k - 1); }
        cd t1;
        larfg2(v1, v2, t1);
        ctx.sync();
        if (tid == 0) {
            if (k > ms) { HH(k, k - 1) = v1; HH(k + 1, k - 1) = czero(); }
            // started below the top of the block (two consecutive small subdiagonals): row ms of the
            // neglected column ms-1 still takes its factor; the fill-in below it is the neglected product
            else if (ms > l) HH(ms, ms - 1) = HH(ms, ms - 1) * (mk(1.0, 0.0) - conj(t1));
        }
        // subdiagonals are general complex numbers here (the multishift chase leaves H(i,i-1)
        // complex), so is t2 = t1 v2
        const cd t2 = t1 * v2;
        for (int j = k + tid; j <= i; j += nt) {
            const cd a = HH(k, j), b = HH(k + 1, j);
            const cd sum = conj(t1) * a + conj(t2) * b;
            HH(k, j) = a - sum;
            HH(k + 1, j) = b - sum * v2;
        }
        ctx.sync();
        const int rmax = (k + 2 < i) ? k + 2 : i;
        for (int j = l + tid; j <= rmax; j += nt) {
            const cd a = HH(j, k), b = HH(j, k + 1);
            const cd sum = t1 * a + t2 * b;
            HH(j, k) = a - sum;
            HH(j, k + 1) = b - sum * conj(v2);
        }
        ctx.sync();
    }
#undef HH
}

// Unblocked multi-bulge chase: every interval works directly on H in global memory.
template <class C>
KB_HD void chase_global(const C& ctx, cd* H, int ld, int l, int i, int ns, const cd* sh, MsRefl* refl) {
#define HH(i_, j_) H[(i_) + (size_t)(j_) * ld]
    const int tid = ctx.tid(), nt = ctx.nthreads();
    const int na = i - l + 1;
    const int T = (na - 1) + 3 * (ns - 1);
    for (int t = 0; t < T; ++t) {
        // active bulges: 0 <= t - 3b <= na - 2
        int b_hi = t / 3;
        if (b_hi > ns - 1) b_hi = ns - 1;
        int b_lo = (t - (na - 2) + 2) / 3;
        if (t - (na - 2) <= 0) b_lo = 0;
        // phase 0: reflectors (one thread per bulge)
        for (int b = b_lo + tid; b <= b_hi; b += nt) {
            const int k = l + t - 3 * b;
            cd v1, v2, t1;
            if (k == l) {
                cd h11s = HH(l, l) - sh[b];
                const cd h21 = HH(l + 1, l);
                const double s = cabs1(h11s) + cabs1(h21);
                if (s == 0.0) { v1 = czero(); v2 = czero(); }
                else { v1 = mk(h11s.x / s, h11s.y / s); v2 = mk(h21.x / s, h21.y / s); }
            } else {
                v1 = HH(k, k - 1);
                v2 = HH(k + 1, k - 1);
            }
            larfg2(v1, v2, t1);
            if (k > l) { HH(k, k - 1) = v1; HH(k + 1, k - 1) = czero(); }
            refl[b].t1 = t1; refl[b].v2 = v2; refl[b].t2 = t1 * v2; refl[b].k = k;
        }
        ctx.sync();
        // phase R: rows k, k+1 ; columns k..i.  Loads of a whole group of bulges are
        // issued before any store (different bulges touch different rows), so the
        // global-memory round trips overlap instead of serialising.
        for (int b0 = b_lo; b0 <= b_hi; b0 += KB_MS_BU) {
            const int span = i - refl[(b0 + KB_MS_BU - 1 <= b_hi) ? b0 + KB_MS_BU - 1 : b_hi].k + 1;
            for (int c0 = 0; c0 < span; c0 += KB_MS_CU * nt) {
                cd va[KB_MS_BU][KB_MS_CU], vb[KB_MS_BU][KB_MS_CU];
#pragma unroll
                for (int u = 0; u < KB_MS_BU; ++u) {
                    const int b = b0 + u;
                    const int k = (b <= b_hi) ? refl[b].k : 0;
#pragma unroll
                    for (int v = 0; v < KB_MS_CU; ++v) {
                        const int j = k + c0 + v * nt + tid;
                        if (b <= b_hi && j <= i) { va[u][v] = HH(k, j); vb[u][v] = HH(k + 1, j); }
                    }
                }
#pragma unroll
                for (int u = 0; u < KB_MS_BU; ++u) {
                    const int b = b0 + u;
                    if (b <= b_hi) {
                        const MsRefl rf = refl[b];
                        const int k = rf.k;
                        const cd ct1 = conj(rf.t1), ct2 = conj(rf.t2);
#pragma unroll
                        for (int v = 0; v < KB_MS_CU; ++v) {
                            const int j = k + c0 + v * nt + tid;
                            if (j <= i) {
                                const cd sum = ct1 * va[u][v] + ct2 * vb[u][v];
                                HH(k, j) = va[u][v] - sum;
                                HH(k + 1, j) = vb[u][v] - sum * rf.v2;
                            }
                        }
                    }
                }
            }
        }
        ctx.sync();
        // phase C: columns k, k+1 ; rows l..min(k+2, i)
        for (int b0 = b_lo; b0 <= b_hi; b0 += KB_MS_BU) {
            const int kmax = refl[b0].k;     // bulge b0 is the lowest of its group
            const int span = ((kmax + 2 < i) ? kmax + 2 : i) - l + 1;
            for (int c0 = 0; c0 < span; c0 += KB_MS_CU * nt) {
                cd va[KB_MS_BU][KB_MS_CU], vb[KB_MS_BU][KB_MS_CU];
#pragma unroll
                for (int u = 0; u < KB_MS_BU; ++u) {
                    const int b = b0 + u;
                    const int k = (b <= b_hi) ? refl[b].k : 0;
                    const int rmax = (k + 2 < i) ? k + 2 : i;
#pragma unroll
                    for (int v = 0; v < KB_MS_CU; ++v) {
                        const int j = l + c0 + v * nt + tid;
                        if (b <= b_hi && j <= rmax) { va[u][v] = HH(j, k); vb[u][v] = HH(j, k + 1); }
                    }
                }
#pragma unroll
                for (int u = 0; u < KB_MS_BU; ++u) {
                    const int b = b0 + u;
                    if (b <= b_hi) {
                        const MsRefl rf = refl[b];
                        const int k = rf.k;
                        const int rmax = (k + 2 < i) ? k + 2 : i;
                        const cd cv2 = conj(rf.v2);
#pragma unroll
                        for (int v = 0; v < KB_MS_CU; ++v) {
                            const int j = l + c0 + v * nt + tid;
                            if (j <= rmax) {
                                const cd sum = rf.t1 * va[u][v] + rf.t2 * vb[u][v];
                                HH(j, k) = va[u][v] - sum;
                                HH(j, k + 1) = vb[u][v] - sum * cv2;
                            }
                        }
                    }
                }
            }
        }
        ctx.sync();
    }
#undef HH
}

// ---------------------------------------------------------------------------------
// Windowed multi-bulge chase.
//
// The unblocked chase moves every row/column of the active block through the CU once per
// interval (n + 3 ns intervals per batch); with H in HBM/L2 that traffic - row accesses are
// strided - is what bounds k_hqr.  Here the intervals are grouped into window steps.  In one
// step the bulge chain moves d = W - 3 ns - 1 rows down inside a W x W diagonal window that
// lives in LDS; every reflector is applied immediately inside the window and only LOGGED for
// the rest of the active block.  After the step the log is replayed on
//   * the right strip  H[ws:we, we:i]   (row operations, one lane per column)
//   * the top strip    H[l:ws, ws:we]   (column operations, one lane per row)
// from LDS tiles: each strip element is read and written once per window step, and the
// replay keeps one operand in a register (bulge-major order: reflectors of different bulges
// act on disjoint rows whenever their time order is swapped, so they commute).
KB_HD int hqr_win_area_elems(int W, int ws) {
    // one LDS image: the W x W window (pitch W+1) or a strip tile of W x wavesize (pitch ws+1)
    const int pitch = (W > ws ? W : ws) + 1;
    return W * pitch;
}
KB_HD int hqr_win_scratch_bytes(int nsmax, int W, int ws) {
    // shift solver area + reflector table (hqr_ms_scratch_bytes), two images, log
    const int logcap = (W + 2) * nsmax;
    return hqr_ms_scratch_bytes(nsmax) + 2 * hqr_win_area_elems(W, ws) * (int)sizeof(cd) +
           logcap * (int)sizeof(MsRefl) + 128;
}

// ---- how the matrix in HBM is addressed.  HPlain: ordinary loads/stores.  HSc1: every access is an
// sc1 (agent-coherent, L1-bypassing, write-through) buffer access - the flavour a workgroup TEAM uses
// for all bytes that cross between its workgroups (MI355X_MICROARCH.md, inter-workgroup visibility:
// sc1 stores + s_waitcnt vmcnt(0) + barrier + sc1 flag / sc1 poll + barrier + sc1 loads).
struct HPlain {
    cd* H;
    int ld;
    int nc = 0;     // columns (needed only where a bounds-checked buffer view is built: kb_hqr2.hpp)
    KB_HD cd get(int r, int c) const { return H[r + (size_t)c * ld]; }
    KB_HD void put(int r, int c, cd v) const { H[r + (size_t)c * ld] = v; }
};
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned int kb_u4 __attribute__((ext_vector_type(4)));
struct HSc1 {
    __amdgpu_buffer_rsrc_t rs;
    int ld;
    __device__ static HSc1 make(cd* H, int ld_, int ncols) {
        HSc1 a;
        a.rs = __builtin_amdgcn_make_buffer_rsrc(H, 0, (int)((size_t)ld_ * ncols * sizeof(cd)), 0x00020000);
        a.ld = ld_;
        return a;
    }
    __device__ cd get(int r, int c) const {
        const kb_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (r + c * ld) * 16, 0, 16);
        cd o;
        o.x = __builtin_bit_cast(double, ((unsigned long long)v.y << 32) | v.x);
        o.y = __builtin_bit_cast(double, ((unsigned long long)v.w << 32) | v.z);
        return o;
    }
    __device__ void put(int r, int c, cd v) const {
        const unsigned long long a = __builtin_bit_cast(unsigned long long, v.x);
        const unsigned long long b = __builtin_bit_cast(unsigned long long, v.y);
        const kb_u4 q = {(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
        __builtin_amdgcn_raw_buffer_store_b128(q, rs, (r + c * ld) * 16, 0, 16);
    }
};
#else
struct HSc1 : HPlain {
    static HSc1 make(cd* H_, int ld_, int) { HSc1 a; a.H = H_; a.ld = ld_; return a; }
};
#endif

// ---- a TEAM = the chase workgroup of a member + one helper workgroup on another CU.
// The chaser keeps everything within 64 of the diagonal to itself (window, the nearest right tile,
// the nearest top tile); the helper replays the logged reflectors on the far tiles.  Records
// (window geometry + reflector log) travel through a small ring in HBM.
struct TeamCtl {                    // 256 B per member, zeroed before the launch
    unsigned published;             // chaser: records published so far (monotonic)
    unsigned done;                  // chaser: no more records will come
    unsigned abort_;                // anybody: a wait timed out - everyone leaves, status is flagged
    unsigned pad0[29];
    unsigned near_done;             // helper: steps g' < near_done have their first round of far right
                                    //         tiles (and every earlier step entirely) complete
    unsigned all_done;              // helper: steps g' < all_done are complete
    unsigned pad1[30];
};
struct TeamRec {                    // 64-byte record header, followed by the log (bulge-major)
    int l, i, ns, na, t0, t1, ws, we, bmin, bmax, nint, g;
    int pad[4];
};
#define KB_TEAM_SLOTS 4
#define KB_TEAM_TOPB 8              // rows above the window that stay with the chase workgroup
// What the chase workgroup keeps next to its window: near_r columns to the right, near_t rows above.
// With 64-lane wavefronts both fit ONE merged tile (lanes 0..55 columns, lanes 56..63 rows).
template <class C> KB_HD int team_near_r() { return (C::WS >= 64) ? C::WS - KB_TEAM_TOPB : C::WS; }
template <class C> KB_HD int team_near_t() { return (C::WS >= 64) ? KB_TEAM_TOPB : C::WS; }
KB_HD int team_rec_bytes(int nsmax, int W) { return (int)sizeof(TeamRec) + (W + 2) * nsmax * (int)sizeof(MsRefl); }

struct WinGeom {
    int l, i, na, ns, t0, t1, nint, ws, we, wlen, bmin, bmax;
};

template <class C>
struct Team;

// Device-side waits: ONE lane polls with sc1 loads (+ s_sleep), result broadcast through LDS.
// Returns false when the team is aborting.  (Host simulation: the helper runs inline, nothing to wait for.)
template <class C>
KB_HD bool team_wait(const C& ctx, volatile unsigned* word, unsigned need, TeamCtl* ctl, int* lds_flag) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (ctx.tid() == 0) {
        int ok = 1;
        const unsigned long long t_start = wall_clock64();
        for (;;) {
            const unsigned v = __hip_atomic_load((unsigned*)word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v >= need) break;
            if (__hip_atomic_load(&ctl->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
            if (wall_clock64() - t_start > 1000000000ull) {         // 10 s at 100 MHz: protocol failure
                __hip_atomic_store(&ctl->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        *lds_flag = ok;
    }
    ctx.sync();
    const bool ok = (*lds_flag != 0);
    ctx.sync();
    return ok;
#else
    (void)ctx; (void)word; (void)need; (void)ctl; (void)lds_flag;
    return true;
#endif
}

// Every wave drains its stores, workgroup barrier, then one lane publishes `value` (sc1 store).
template <class C>
KB_HD void team_signal(const C& ctx, unsigned* word, unsigned value) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ctx.sync();
    if (ctx.tid() == 0) __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    (void)ctx;
    *word = value;
#endif
}

// Replay of the reflector log on strip tiles staged in LDS (two tiles per round: wavefront w works on
// tile slot w % 2).  Tiles are numbered by distance from the window:
//   right tile j : rows ws..we-1, columns we + 64 j ..          (row operations, one lane per column)
//   top tile j   : columns ws..we-1, rows ws - 64 (j+1) .. ws - 64 j - 1, not above l (column operations,
//                  one lane per row)
// Processes right tiles [rlo, rhi) then top tiles [tlo, thi); after_round(r) runs after each round.
// roff / toff shift the tile grids away from the window (the helper's tiles start beyond the chaser's
// near region).  MERGED: one single tile holding the chaser's near region (team_near_r columns on the
// low lanes, team_near_t rows on the high lanes), replayed by all wavefronts.
template <int MERGED, class C, class ACC, class F>
KB_HD void strip_tiles(const C& ctx, const ACC& A, const WinGeom& G, const MsRefl* logv, cd* Hw, cd* Tile,
                       int rlo, int rhi, int tlo, int thi, int roff, int toff, MsStats* stats, F&& after_round) {
    const int tid = ctx.tid();
    const int TP = C::WS + 1;
    const int l = G.l, i = G.i, na = G.na, t0 = G.t0, nint = G.nint, ws = G.ws, we = G.we, wlen = G.wlen;
    const int bmin = G.bmin, bmax = G.bmax;
    // With several wavefronts per tile they split its bulges (and its rows for the copies); a bulge on one
    // wavefront may follow a bulge of another by two intervals on the same rows, hence one workgroup
    // barrier per interval.  Every wavefront runs the same trip counts (barrier safety).
    const int ntb = (!MERGED && ctx.nwaves() >= 2) ? 2 : 1;
    // wavefronts per tile: 1, 2, 4 or 8 (each takes KB_MS_RG / nhalf bulges of every group)
    const int nhalf = (ctx.nwaves() >= 8 * ntb) ? 8 : (ctx.nwaves() >= 4 * ntb) ? 4 : ((ctx.nwaves() >= 2 * ntb) ? 2 : 1);
    const int slot = ctx.wave() % ntb, half = ctx.wave() / ntb;
    const bool worker = ctx.wave() < ntb * nhalf;
    const int nr = (rhi > rlo) ? rhi - rlo : 0, ntp = (thi > tlo) ? thi - tlo : 0;
    const int ntiles = MERGED ? ((nr + ntp > 0) ? 1 : 0) : nr + ntp;
    const int near_r = team_near_r<C>(), near_t = team_near_t<C>();
    cd* Tl = (slot == 0) ? Hw : Tile;
    const int lane = ctx.lane();
    const int ngrp = (bmax - bmin) / KB_MS_RG + 1;
    for (int round = 0; round * ntb < ntiles; ++round) {
        const int tile = round * ntb + slot;
        const bool right = MERGED ? (lane < near_r) : (tile < nr);
        const int q = MERGED ? (right ? we + lane : ws - near_t + (lane - near_r))
                             : (right ? we + roff + (rlo + tile) * C::WS : ws - toff - (tlo + tile - nr + 1) * C::WS) + lane;
        const bool live = worker && tile < ntiles && (right ? (q <= i && nr > 0) : (q >= l && q < ws && ntp > 0));
        const double sg = right ? -1.0 : 1.0;                // right strip uses conj(t1), conj(t2), v2
        const long long c_t0 = KB_CLOCK();
        if (live) {
            // this wavefront's share of the rows; eight global loads in flight per lane
            int p = half;
            for (; p + 7 * nhalf < wlen; p += 8 * nhalf) {
                cd v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = right ? A.get(ws + p + u * nhalf, q) : A.get(q, ws + p + u * nhalf);
#pragma unroll
                for (int u = 0; u < 8; ++u) Tl[(p + u * nhalf) * TP + lane] = v[u];
            }
            for (; p < wlen; p += nhalf) Tl[p * TP + lane] = right ? A.get(ws + p, q) : A.get(q, ws + p);
        }
        ctx.sync();
        const long long c_t1 = KB_CLOCK();
        auto replay = [&](auto nu_tag) {
            constexpr int NU = decltype(nu_tag)::value;      // bulges handled by this wavefront per group
            for (int g = 0; g < ngrp; ++g) {
                const int g0 = g * KB_MS_RG;
                int tb[NU], te[NU], pp[NU];
                cd carry[NU];
                int tt_lo = nint, tt_hi = 0;
                // interval range of the whole group (uniform over the workgroup)
                for (int u = 0; u < KB_MS_RG; ++u) {
                    const int bb_ = bmin + g0 + u;
                    int b0_ = 3 * bb_ - t0;
                    if (b0_ < 0) b0_ = 0;
                    int e0_ = 3 * bb_ + (na - 2) - t0 + 1;
                    if (e0_ > nint) e0_ = nint;
                    if (g0 + u <= bmax - bmin && e0_ > b0_) {
                        tt_lo = b0_ < tt_lo ? b0_ : tt_lo;
                        tt_hi = e0_ > tt_hi ? e0_ : tt_hi;
                    }
                }
                // the bulges this wavefront replays: local u -> group index
                const int u_base = half * NU;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int gu = u_base + u;
                    const int bb_ = bmin + g0 + gu;
                    int b0_ = 3 * bb_ - t0;
                    if (b0_ < 0) b0_ = 0;
                    int e0_ = 3 * bb_ + (na - 2) - t0 + 1;
                    if (e0_ > nint) e0_ = nint;
                    if (g0 + gu > bmax - bmin) { b0_ = 0; e0_ = 0; }
                    tb[u] = b0_; te[u] = e0_;
                    pp[u] = (l + (t0 + b0_) - 3 * bb_) - ws;
                    carry[u] = czero();
                }
                for (int tt = tt_lo; tt < tt_hi; ++tt) {
                    if (live) {
                        // LDS reads of the interval, then the arithmetic, then the writes
                        cd nbv[NU], c1v[NU], c2v[NU], c3v[NU], outv[NU];
#pragma unroll
                        for (int u = 0; u < NU; ++u) {
                            if (tt >= tb[u] && tt < te[u]) {
                                const int p = pp[u] + (tt - tb[u]);
                                const MsRefl* lg = logv + (g0 + u_base + u) * nint + tt;
                                if (tt == tb[u]) carry[u] = Tl[p * TP + lane];
                                nbv[u] = Tl[(p + 1) * TP + lane];
                                const cd t1 = lg->t1, t2 = lg->t2, v2 = lg->v2;
                                c1v[u] = mk(t1.x, sg * t1.y);
                                c2v[u] = mk(t2.x, sg * t2.y);
                                c3v[u] = mk(v2.x, -sg * v2.y);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < NU; ++u) {
                            if (tt >= tb[u] && tt < te[u]) {
                                const cd sum = c1v[u] * carry[u] + c2v[u] * nbv[u];
                                outv[u] = carry[u] - sum;
                                carry[u] = nbv[u] - sum * c3v[u];
                            }
                        }
#pragma unroll
                        for (int u = 0; u < NU; ++u) {
                            if (tt >= tb[u] && tt < te[u]) {
                                const int p = pp[u] + (tt - tb[u]);
                                Tl[p * TP + lane] = outv[u];
                                if (tt == te[u] - 1) Tl[(p + 1) * TP + lane] = carry[u];
                            }
                        }
                    }
                    if (nhalf > 1) ctx.sync();
                }
            }
        };
        if (nhalf == 8) replay(KbInt<KB_MS_RG / 8>{});
        else if (nhalf == 4) replay(KbInt<KB_MS_RG / 4>{});
        else if (nhalf == 2) replay(KbInt<KB_MS_RG / 2>{});
        else replay(KbInt<KB_MS_RG>{});
        ctx.sync();
        const long long c_t2 = KB_CLOCK();
        if (live) {
            for (int p = half; p < wlen; p += nhalf) {
                if (right) A.put(ws + p, q, Tl[p * TP + lane]);
                else A.put(q, ws + p, Tl[p * TP + lane]);
            }
        }
        ctx.sync();
        if (stats && tid == 0) {
            const long long c_t3 = KB_CLOCK();
            stats->cyc_tload += c_t1 - c_t0; stats->cyc_treplay += c_t2 - c_t1; stats->cyc_tstore += c_t3 - c_t2;
            stats->ntiles++;
        }
        after_round(round);
    }
}

KB_HD void win_tile_counts(const WinGeom& G, int wsz, int roff, int toff, int& tiles_r, int& tiles_t) {
    int nright = ((G.we <= G.i) ? G.i - G.we + 1 : 0) - roff;
    int ntop = G.ws - G.l - toff;
    if (nright < 0) nright = 0;
    if (ntop < 0) ntop = 0;
    tiles_r = (nright + wsz - 1) / wsz;
    tiles_t = (ntop + wsz - 1) / wsz;
}

// LDS carve shared by the chase workgroup and the helper workgroup (after S, sh, refl, sinfo).
struct WinLds {
    cd* Hw;
    cd* Tile;
    MsRefl* logv;
    int* flag;
};
template <class C>
KB_HD WinLds win_lds(const C& ctx, int W, int nsmax) {
    cd* S = reinterpret_cast<cd*>(ctx.scratch());
    cd* sh = S + nsmax * nsmax;
    MsRefl* refl = reinterpret_cast<MsRefl*>(sh + nsmax);
    int* sinfo = reinterpret_cast<int*>(refl + nsmax);
    char* base = reinterpret_cast<char*>(refl + nsmax) + 64;   // past refl[nsmax] and the info words
    const int area = hqr_win_area_elems(W, C::WS);
    WinLds L;
    L.Hw = reinterpret_cast<cd*>(base);
    L.Tile = L.Hw + area;
    L.logv = reinterpret_cast<MsRefl*>(L.Tile + area);
    L.flag = sinfo + 4;
    return L;
}

template <class C>
struct Team {
    TeamCtl* ctl;
    char* ring;           // KB_TEAM_SLOTS records of rec_bytes each
    int rec_bytes;
    unsigned g;           // chaser: global step counter
    unsigned g_batch;     // chaser: first step of the current batch
    int failed;           // a wait was aborted
    HSc1 A;               // the matrix, team flavour
    int W, nsmax;
};

// The helper's share of one record: far right tiles (first round reported through near_done), then far
// top tiles.  Also the body of the host simulation's inline helper.
template <class C>
KB_HD void team_helper_record(const C& ctx, Team<C>& tm, const WinGeom& G, unsigned g, const WinLds& L) {
    const int roff = team_near_r<C>(), toff = team_near_t<C>();
    int tiles_r, tiles_t;
    win_tile_counts(G, C::WS, roff, toff, tiles_r, tiles_t);
    bool near_sent = false;
    if (tiles_r > 0) {
        strip_tiles<0>(ctx, tm.A, G, L.logv, L.Hw, L.Tile, 0, tiles_r, 0, 0, roff, toff, nullptr, [&](int round) {
            if (round == 0) { team_signal(ctx, &tm.ctl->near_done, g + 1); near_sent = true; }
        });
    }
    if (!near_sent) team_signal(ctx, &tm.ctl->near_done, g + 1);
    if (tiles_t > 0) strip_tiles<0>(ctx, tm.A, G, L.logv, L.Hw, L.Tile, 0, 0, 0, tiles_t, roff, toff, nullptr, [](int) {});
    team_signal(ctx, &tm.ctl->all_done, g + 1);
}

// Helper workgroup main loop: consume records until the chaser is done.
template <class C>
KB_HD void team_helper_main(const C& ctx, Team<C>& tm) {
#if defined(__HIP_DEVICE_COMPILE__)
    const WinLds L = win_lds(ctx, tm.W, tm.nsmax);
    const int tid = ctx.tid(), nt = ctx.nthreads();
    for (unsigned g = 0;; ++g) {
        // wait for record g or for the end
        if (tid == 0) {
            int st = 0;                         // 1: record ready, 2: finished, 3: abort
            const unsigned long long t_start = wall_clock64();
            for (;;) {
                if (__hip_atomic_load(&tm.ctl->published, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > g) { st = 1; break; }
                if (__hip_atomic_load(&tm.ctl->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    // done is stored after the last publish: look once more
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
        // record -> registers / LDS (sc1 loads)
        const char* rec = tm.ring + (size_t)(g % KB_TEAM_SLOTS) * tm.rec_bytes;
        __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(rec), 0, tm.rec_bytes, 0x00020000);
        WinGeom G;
        {
            const kb_u4 h0 = __builtin_amdgcn_raw_buffer_load_b128(rr, 0, 0, 16);
            const kb_u4 h1 = __builtin_amdgcn_raw_buffer_load_b128(rr, 16, 0, 16);
            const kb_u4 h2 = __builtin_amdgcn_raw_buffer_load_b128(rr, 32, 0, 16);
            G.l = (int)h0.x; G.i = (int)h0.y; G.ns = (int)h0.z; G.na = (int)h0.w;
            G.t0 = (int)h1.x; G.t1 = (int)h1.y; G.ws = (int)h1.z; G.we = (int)h1.w;
            G.bmin = (int)h2.x; G.bmax = (int)h2.y; G.nint = (int)h2.z;
            G.wlen = G.we - G.ws;
        }
        const int nq = (G.bmax - G.bmin + 1) * G.nint * 4;          // 16-byte words of the log
        kb_u4* dst = reinterpret_cast<kb_u4*>(L.logv);
        for (int idx = tid; idx < nq; idx += nt)
            dst[idx] = __builtin_amdgcn_raw_buffer_load_b128(rr, (int)sizeof(TeamRec) + idx * 16, 0, 16);
        ctx.sync();
        team_helper_record(ctx, tm, G, g, L);
    }
#else
    (void)ctx; (void)tm;
#endif
}

template <class C, class ACC>
KB_HD void chase_windowed(const C& ctx, const ACC& A, int l, int i, int ns, const cd* sh, MsRefl* refl,
                          int W, int nsmax, MsStats* stats, Team<C>* team = nullptr) {
#define HW(i_, j_) Hw[((i_) - ws) + ((j_) - ws) * WP]
    const int tid = ctx.tid(), nt = ctx.nthreads();
    const int na = i - l + 1;
    const int WP = W + 1;                                   // padded pitch of the LDS images
    const WinLds L = win_lds(ctx, W, nsmax);
    cd* Hw = L.Hw;
    cd* Tile = L.Tile;
    MsRefl* logv = L.logv;
    const int T = (na - 1) + 3 * (ns - 1);
    int t0 = 0;
    if (team) team->g_batch = team->g;
    while (t0 < T) {
        // ---- window of this step: first row = row above the topmost active bulge
        int bh0 = t0 / 3;
        if (bh0 > ns - 1) bh0 = ns - 1;
        const int kmin0 = l + t0 - 3 * bh0;
        int ws = kmin0 - 1;
        if (ws < l) ws = l;
        int we = ws + W;                                    // exclusive
        if (we > i + 1) we = i + 1;
        // ---- how many intervals fit: lowest bulge must keep k+2 inside the window
        int t1 = t0;
        for (; t1 < T && t1 - t0 < W; ++t1) {             // at most W intervals per step (log capacity)
            int blo = 0;
            if (t1 - (na - 2) > 0) blo = (t1 - (na - 2) + 2) / 3;
            const int kmax = l + t1 - 3 * blo;
            const int reach = (kmax + 2 < i) ? kmax + 2 : i;
            if (reach > we - 1) break;
            int bhi = t1 / 3;
            if (bhi > ns - 1) bhi = ns - 1;
            const int kmin = l + t1 - 3 * bhi;
            if (kmin > l && kmin - 1 < ws) break;           // cannot happen (kmin never decreases below ws+1)
        }
        if (t1 == t0) t1 = t0 + 1;     // unreachable for W >= 3 ns + 8; never spin
        const int nint = t1 - t0;
        int bmin = 0;
        if (t0 - (na - 2) > 0) bmin = (t0 - (na - 2) + 2) / 3;
        int bmax = (t1 - 1) / 3;
        if (bmax > ns - 1) bmax = ns - 1;
        // log, bulge-major: entry (b - bmin) * nint + (t - t0); k < 0 marks "not active"
        for (int idx = tid; idx < (bmax - bmin + 1) * nint; idx += nt) logv[idx].k = -1;
        // ---- (a) load the diagonal window
        const long long c_a = KB_CLOCK();
        const int wlen = we - ws;
        for (int idx = tid; idx < wlen * wlen; idx += nt) {
            const int r = idx % wlen, c = idx / wlen;
            Hw[r + c * WP] = A.get(ws + r, ws + c);
        }
        ctx.sync();
        const long long c_b = KB_CLOCK();
        // ---- (b) chase inside the window, logging the reflectors.  Bulge positions are arithmetic
        // (k = l + t - 3 b), so the reflector-table read and the matrix reads are independent LDS loads;
        // the item -> (bulge, offset) split of the first pass is hoisted out of the interval loop.
        const int q0 = tid / W, r0 = tid % W;
        const int cmax = (i < we - 1) ? i : we - 1;
        for (int t = t0; t < t1; ++t) {
            int b_hi = t / 3;
            if (b_hi > ns - 1) b_hi = ns - 1;
            int b_lo = 0;
            if (t - (na - 2) > 0) b_lo = (t - (na - 2) + 2) / 3;
            for (int b = b_lo + tid; b <= b_hi; b += nt) {
                const int k = l + t - 3 * b;
                cd v1, v2, t1c;
                if (k == l) {
                    cd h11s = HW(l, l) - sh[b];
                    const cd h21 = HW(l + 1, l);
                    const double sc = cabs1(h11s) + cabs1(h21);
                    if (sc == 0.0) { v1 = czero(); v2 = czero(); }
                    else { v1 = mk(h11s.x / sc, h11s.y / sc); v2 = mk(h21.x / sc, h21.y / sc); }
                } else {
                    v1 = HW(k, k - 1);
                    v2 = HW(k + 1, k - 1);
                }
                larfg2(v1, v2, t1c);
                if (k > l) { HW(k, k - 1) = v1; HW(k + 1, k - 1) = czero(); }
                MsRefl rf;
                rf.t1 = t1c; rf.v2 = v2; rf.t2 = t1c * v2; rf.k = k; rf.pad = b;
                refl[b] = rf;
                logv[(b - bmin) * nint + (t - t0)] = rf;
            }
            ctx.sync();
            const int nb = b_hi - b_lo + 1;
            // rows k, k+1 ; columns k..min(i, we-1)
            for (int idx = tid; idx < nb * W; idx += nt) {
                int q = q0, o = r0;
                if (idx != tid) { q = idx / W; o = idx % W; }
                const int b = b_lo + q;
                const int k = l + t - 3 * b;
                const int j = k + o;
                if (j <= cmax) {
                    const cd t1c = refl[b].t1, t2c = refl[b].t2, v2c = refl[b].v2;
                    const cd a = HW(k, j), bb = HW(k + 1, j);
                    const cd sum = conj(t1c) * a + conj(t2c) * bb;
                    HW(k, j) = a - sum;
                    HW(k + 1, j) = bb - sum * v2c;
                }
            }
            ctx.sync();
            // columns k, k+1 ; rows ws..min(k+2, i)
            for (int idx = tid; idx < nb * W; idx += nt) {
                int q = q0, o = r0;
                if (idx != tid) { q = idx / W; o = idx % W; }
                const int b = b_lo + q;
                const int k = l + t - 3 * b;
                const int rmax = (k + 2 < i) ? k + 2 : i;
                const int r = ws + o;
                if (r <= rmax) {
                    const cd t1c = refl[b].t1, t2c = refl[b].t2, v2c = conj(refl[b].v2);
                    const cd a = HW(r, k), bb = HW(r, k + 1);
                    const cd sum = t1c * a + t2c * bb;
                    HW(r, k) = a - sum;
                    HW(r, k + 1) = bb - sum * v2c;
                }
            }
            ctx.sync();
        }
        const long long c_c = KB_CLOCK();
        WinGeom G;
        G.l = l; G.i = i; G.na = na; G.ns = ns; G.t0 = t0; G.t1 = t1; G.nint = nint; G.ws = ws; G.we = we;
        G.wlen = wlen; G.bmin = bmin; G.bmax = bmax;
        if (team) {
            // ---- publish the record (geometry + log) first: the helper's far tiles do not depend on
            // the window image.  Slot reuse is safe: near_done >= g - 1 (waited for below at step g - 1)
            // implies all_done >= g - 2.
            Team<C>& tm = *team;
            char* rec = tm.ring + (size_t)(tm.g % KB_TEAM_SLOTS) * tm.rec_bytes;
#if defined(__HIP_DEVICE_COMPILE__)
            __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(rec, 0, tm.rec_bytes, 0x00020000);
            if (tid < 3) {
                kb_u4 h;
                if (tid == 0) h = kb_u4{(unsigned)l, (unsigned)i, (unsigned)ns, (unsigned)na};
                else if (tid == 1) h = kb_u4{(unsigned)t0, (unsigned)t1, (unsigned)ws, (unsigned)we};
                else h = kb_u4{(unsigned)bmin, (unsigned)bmax, (unsigned)nint, tm.g};
                __builtin_amdgcn_raw_buffer_store_b128(h, rr, tid * 16, 0, 16);
            }
            const int nq = (bmax - bmin + 1) * nint * 4;
            const kb_u4* src = reinterpret_cast<const kb_u4*>(logv);
            for (int idx = tid; idx < nq; idx += nt)
                __builtin_amdgcn_raw_buffer_store_b128(src[idx], rr, (int)sizeof(TeamRec) + idx * 16, 0, 16);
#else
            (void)rec;
#endif
            team_signal(ctx, &tm.ctl->published, tm.g + 1);
        }
        // ---- (c) store the window back
        for (int idx = tid; idx < wlen * wlen; idx += nt) {
            const int r = idx % wlen, c = idx / wlen;
            A.put(ws + r, ws + c, Hw[r + c * WP]);
        }
        ctx.sync();
        const long long c_d = KB_CLOCK();
        int tiles_r, tiles_t;
        win_tile_counts(G, C::WS, 0, 0, tiles_r, tiles_t);
        if (!team) {
            // ---- (d) right strip, (e) top strip: every tile
            strip_tiles<0>(ctx, A, G, logv, Hw, Tile, 0, tiles_r, 0, tiles_t, 0, 0, stats, [](int) {});
        } else {
            Team<C>& tm = *team;
#if !defined(__HIP_DEVICE_COMPILE__)
            // host simulation: the helper's share of this record runs inline, in the order the protocol
            // allows at the latest (before the chaser's own tiles of the same step would be too early
            // for nothing: the regions are disjoint)
            team_helper_record(ctx, tm, G, tm.g, L);
#endif
            // the nearest right / top tile of this step touch elements the helper wrote at step g - 1
            // (its first far right tile) and earlier: wait for those; at the first step of a batch for
            // everything before (top strips of the previous batch reach down to this window's columns)
            bool ok = true;
            if (tm.g > 0 && !tm.failed) {
                if (tm.g == tm.g_batch) ok = team_wait(ctx, &tm.ctl->all_done, tm.g, tm.ctl, L.flag);
                else ok = team_wait(ctx, &tm.ctl->near_done, tm.g, tm.ctl, L.flag);
            }
            if (!ok) tm.failed = 1;
            // ---- (d)+(e) the near region: one merged tile (64-lane wavefronts) or one tile each
            if (!tm.failed) {
                if (C::WS >= 64)
                    strip_tiles<1>(ctx, A, G, logv, Hw, Tile, 0, tiles_r > 0 ? 1 : 0, 0, tiles_t > 0 ? 1 : 0, 0, 0, stats, [](int) {});
                else
                    strip_tiles<0>(ctx, A, G, logv, Hw, Tile, 0, tiles_r > 0 ? 1 : 0, 0, tiles_t > 0 ? 1 : 0, 0, 0, stats, [](int) {});
            }
            tm.g++;
        }
        ctx.sync();
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

// All eigenvalues of a small unreduced upper Hessenberg matrix T (n <= wavefront size) by simultaneous
// Ehrlich-Aberth iteration on the characteristic polynomial, ONE LANE PER ROOT: p(z) and p'(z) come from
// Hyman's recurrence (the left null-vector recurrence of T - z I, O(n^2) per root and evaluation, backward
// stable for Hessenberg matrices), roots start at the diagonal.  Where the one-wavefront QR iteration
// (hqr_eigvals) is a chain of ~3 n^2 dependent rotation steps, this is a handful of iterations that all lanes
// take in parallel: the shift computation of the multishift iteration drops from ~140 k to ~20 k cycles.
// The roots are accurate to a few ulp of |z| (they are used as shifts: a shift error of 1e-6 already costs
// half as many sweeps again).  Returns false (caller falls back to hqr_eigvals) on a zero subdiagonal, a
// non-finite value or no convergence within maxit iterations.
//   wc: wavefront context; T column-major (ldt), read only; z: n roots out (LDS);
//   U, D: n*n each (LDS; recurrence vectors, [i * n + root]);  zw: 2 n (LDS).
template <class WC>
KB_HD bool aberth_eigs(const WC& wc, int n, const cd* T, int ldt, cd* z, cd* U, cd* D, cd* zw, int maxit) {
#define TT(i_, j_) T[(i_) + (j_) * ldt]
    const int lane = wc.lane();
    constexpr int WS = WC::WS;
    cd* rinv = zw + n;
    double sc = 0.0;
    int bad = 0;
    for (int idx = lane; idx < n * n; idx += WS) {
        const int r = idx % n, c = idx / n;
        if (r <= c + 1) sc = fmax(sc, cabs1(TT(r, c)));
    }
    sc = wc.wave_max(sc);
    for (int r = lane; r < n; r += WS) {
        z[r] = TT(r, r);
        if (r < n - 1) {
            const cd h = TT(r + 1, r);
            if (is_zero(h)) bad = 1;
            else rinv[r] = cdiv(mk(1.0, 0.0), h);
        }
    }
    wc.sync();
    // coincident starting points would divide by zero in the Aberth sum: spread them
    for (int r = lane; r < n; r += WS) {
        cd zr = z[r];
        for (int k = 0; k < r; ++k)
            if (cabs1(zr - z[k]) <= 1e-8 * sc) {
                const double a = 1e-4 * sc * (double)(r + 1);
                zr = zr + mk(a * (1.0 - 0.125 * k), a * 0.0625 * (k + 1));
            }
        zw[r] = zr;
    }
    wc.sync();
    for (int r = lane; r < n; r += WS) { z[r] = zw[r]; zw[r] = czero(); }   // zw[r].x becomes the "converged" flag
    wc.sync();
    if (wc.wave_max(bad) != 0 || !(sc > 0.0)) return false;
    bool conv_all = false;
    for (int it = 0; it < maxit && !conv_all; ++it) {
        int open_ = 0;
        for (int r = lane; r < n; r += WS) {
            const cd zr = z[r];
            // Hyman: u_0 = 1; u_{j+1} = -(sum_{i<=j} u_i T(i,j) - z u_j) / T(j+1,j); p = the last sum
            cd* u = U + r;
            cd* d = D + r;
            u[0] = mk(1.0, 0.0);
            d[0] = czero();
            cd acc = czero(), dacc = czero();
            for (int j = 0; j < n; ++j) {
                acc = czero();
                dacc = czero();
                for (int i = 0; i <= j; ++i) {
                    const cd t = TT(i, j);
                    cfma(acc, u[i * n], t);
                    cfma(dacc, d[i * n], t);
                }
                const cd uj = u[j * n], dj = d[j * n];
                cfma(acc, -zr, uj);
                cfma(dacc, -zr, dj);
                dacc = dacc - uj;
                if (j < n - 1) {
                    u[(j + 1) * n] = -(acc * rinv[j]);
                    d[(j + 1) * n] = -(dacc * rinv[j]);
                }
            }
            cd dz = czero();
            const bool frozen = zw[r].x != 0.0;         // zw[r].x: 1 once the root has converged
            if (!frozen) {
                if (is_zero(acc)) dz = czero();
                else {
                    const cd nw = cdiv(acc, dacc);       // Newton correction p / p'
                    cd sum = czero();
                    for (int k = 0; k < n; ++k)
                        if (k != r) sum = sum + cdiv(mk(1.0, 0.0), zr - z[k]);
                    dz = cdiv(nw, mk(1.0, 0.0) - nw * sum);
                }
                if (!(cabs1(dz) < 1e300)) { bad = 1; dz = czero(); }
                if (cabs1(dz) <= 4.0 * KB_ULP * fmax(cabs1(zr), 0.015625 * sc)) zw[r].x = 1.0;
                else open_ = 1;
            }
            zw[r].y = 0.0;
            U[r] = zr - dz;                              // u_0 slot reused as the new root (rewritten next pass)
        }
        wc.sync();
        for (int r = lane; r < n; r += WS) z[r] = U[r];
        wc.sync();
        conv_all = wc.wave_max(open_) == 0;
        if (wc.wave_max(bad) != 0) return false;
    }
    return conv_all;
#undef TT
}

// The same iteration for exactly NR roots with the recurrence vectors in registers (fully unrolled: static
// indices), T read through LDS broadcasts only - the common case of the multishift iteration (ns = NR = 8).
// 1 / (z_r - z_k) is formed as conj / |.|^2 (the differences of distinct roots of a matrix of norm ~sc are far
// from the under/overflow thresholds; a non-finite value ends in the fallback like everything else).
template <int NR, class WC>
KB_HD bool aberth_eigs_reg(const WC& wc, const cd* __restrict__ T, int ldt, cd* z, cd* zw, int maxit, int* iters) {
#define TT(i_, j_) T[(i_) + (j_) * ldt]
    const int lane = wc.lane();
    constexpr int WS = WC::WS;
    constexpr int n = NR;
    cd* rinv = zw + n;
    double sc = 0.0;
    int bad = 0;
    for (int idx = lane; idx < n * n; idx += WS) {
        const int r = idx % n, c = idx / n;
        if (r <= c + 1) sc = fmax(sc, cabs1(TT(r, c)));
    }
    sc = wc.wave_max(sc);
    for (int r = lane; r < n; r += WS) {
        z[r] = TT(r, r);
        if (r < n - 1) {
            const cd h = TT(r + 1, r);
            if (is_zero(h)) bad = 1;
            else rinv[r] = cdiv(mk(1.0, 0.0), h);
        }
    }
    wc.sync();
    for (int r = lane; r < n; r += WS) {
        cd zr = z[r];
        for (int k = 0; k < r; ++k)
            if (cabs1(zr - z[k]) <= 1e-8 * sc) {
                const double a = 1e-4 * sc * (double)(r + 1);
                zr = zr + mk(a * (1.0 - 0.125 * k), a * 0.0625 * (k + 1));
            }
        zw[r] = zr;
    }
    wc.sync();
    for (int r = lane; r < n; r += WS) { z[r] = zw[r]; zw[r] = czero(); }
    wc.sync();
    if (wc.wave_max(bad) != 0 || !(sc > 0.0)) return false;
    bool conv_all = false;
    int it = 0;
    for (; it < maxit && !conv_all; ++it) {
        int open_ = 0;
        for (int r = lane; r < n; r += WS) {
            const cd zr = z[r];
            cd u[NR], d[NR];
            u[0] = mk(1.0, 0.0);
            d[0] = czero();
            cd acc = czero(), dacc = czero();
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                // two partial sums per recurrence: shorter dependent chains
                cd a0 = czero(), a1 = czero(), b0 = czero(), b1 = czero();
#pragma unroll
                for (int i = 0; i <= j; ++i) {
                    const cd t = TT(i, j);
                    if (i & 1) { cfma(a1, u[i], t); cfma(b1, d[i], t); }
                    else { cfma(a0, u[i], t); cfma(b0, d[i], t); }
                }
                acc = a0 + a1;
                dacc = b0 + b1;
                cfma(acc, -zr, u[j]);
                cfma(dacc, -zr, d[j]);
                dacc = dacc - u[j];
                if (j < NR - 1) {
                    const cd ri = rinv[j];
                    u[j + 1] = -(acc * ri);
                    d[j + 1] = -(dacc * ri);
                }
            }
            cd dz = czero();
            const bool frozen = zw[r].x != 0.0;
            if (!frozen) {
                if (!is_zero(acc)) {
                    const cd nw = cdiv(acc, dacc);
                    cd sum = czero();
#pragma unroll
                    for (int k = 0; k < NR; ++k) {
                        const cd df = zr - z[k];
                        const double q = 1.0 / abs2(df);
                        if (k != r) sum = sum + mk(df.x * q, -df.y * q);
                    }
                    dz = cdiv(nw, mk(1.0, 0.0) - nw * sum);
                }
                if (!(cabs1(dz) < 1e300)) { bad = 1; dz = czero(); }
                if (cabs1(dz) <= 4.0 * KB_ULP * fmax(cabs1(zr), 0.015625 * sc)) zw[r].x = 1.0;
                else open_ = 1;
            }
            zw[r].y = 0.0;
            rinv[n + r] = zr - dz;
        }
        wc.sync();
        for (int r = lane; r < n; r += WS) z[r] = rinv[n + r];
        wc.sync();
        conv_all = wc.wave_max(open_) == 0;
        if (wc.wave_max(bad) != 0) return false;
    }
    if (iters) *iters = it;
    return conv_all;
#undef TT
}

template <class C>
KB_HD void hqr_eigvals_ms(const C& ctx, int n, cd* H, int ld, cd* w, int* info, int nsmax,
                          MsStats* stats = nullptr, int win_w = 0, Team<C>* team = nullptr, bool aberth = true) {
#define HH(i_, j_) H[(i_) + (size_t)(j_) * ld]
    const double ulp = KB_ULP;
    const double smlnum = KB_SAFMIN * ((double)n / ulp);
    const int tid = ctx.tid(), nt = ctx.nthreads();
    if (nsmax > KB_MS_NSMAX) nsmax = KB_MS_NSMAX;
    cd* S = reinterpret_cast<cd*>(ctx.scratch());
    cd* sh = S + nsmax * nsmax;
    MsRefl* refl = reinterpret_cast<MsRefl*>(sh + nsmax);
    int* sinfo = reinterpret_cast<int*>(refl + nsmax);
    int fail = 0;
    bool bail = false;
    // workspace of the Aberth shift solver: the window image, idle between two chases
    cd* aws = (win_w > 0 && aberth) ? win_lds(ctx, win_w, nsmax).Hw : nullptr;
    if (n == 1) {
        if (tid == 0) { w[0] = HH(0, 0); *info = 0; }
        ctx.sync();
        if (team) team_signal(ctx, &team->ctl->done, 1u);
        return;
    }
    // (subdiagonals stay general complex numbers throughout: no realness is maintained)
    ctx.sync();
    const int itmax = 30 * (n > 10 ? n : 10);
    int kdefl = 0;
    int i = n - 1;
    const long long c_total0 = KB_CLOCK();
    while (i >= 0) {
        int l = 0;
        int done = 0;   // 1: H(i,i) converged, 2: 2x2 block solved
        for (int its = 0; its <= itmax; ++its) {
            const long long c_scan0 = KB_CLOCK();
            // ---- deflation scan: largest k in (l, i] with a negligible subdiagonal
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
            if (na < KB_MS_MIN || nsmax < 2) {
                const long long c0 = KB_CLOCK();
                single_shift_sweep(ctx, H, ld, l, i, kdefl);
                if (stats && tid == 0) { stats->single_sweeps++; stats->cyc_single += KB_CLOCK() - c0; }
            } else {
                const long long c_sh0 = KB_CLOCK();
                int ns = na / 3;
                if (ns > nsmax) ns = nsmax;
                if (ns < 2) ns = 2;
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
                    if (ctx.wave() == 0) {
                        WaveCtx<C> wc{ctx, nullptr, 0};
                        bool ok = false;
                        if (aws && ns >= 3 && ns <= C::WS * 64) {
                            int iters = 0;
                            if (ns == 8) ok = aberth_eigs_reg<8>(wc, S, ns, sh, aws, 40, &iters);
                            else ok = aberth_eigs(wc, ns, S, ns, sh, aws, aws + ns * ns, aws + 2 * ns * ns, 40);
                            if (stats && tid == 0) { stats->ab_calls++; stats->ab_iters += iters; if (!ok) stats->ab_fail++; }
                        }
                        if (!ok) hqr_eigvals(wc, ns, S, ns, sh, sinfo);
                    }
                }
                ctx.sync();
                if (stats && tid == 0) stats->cyc_shift += KB_CLOCK() - c_sh0;
                // ---- pipelined chase of ns bulges, 3 rows apart
                const int T = (na - 1) + 3 * (ns - 1);
                if (win_w >= 3 * ns + 8) {
                    if (team) {
                        chase_windowed(ctx, team->A, l, i, ns, sh, refl, win_w, nsmax, stats, team);
#if defined(__HIP_DEVICE_COMPILE__)
                        // the scan / shift / small-block code reads the band with plain loads: drop
                        // whatever this CU's L1 still holds of it (the window was stored sc1)
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
                        if (team->failed) { fail = 1; bail = true; }
                    } else {
                        chase_windowed(ctx, HPlain{H, ld}, l, i, ns, sh, refl, win_w, nsmax, stats);
                    }
                } else
                    chase_global(ctx, H, ld, l, i, ns, sh, refl);
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
