// Building blocks of the multishift QR iteration on an upper Hessenberg matrix (kb_hqr2.hpp; part of the zgeev
// replacement, reference kbdm.py:192): wavefront context, instrumentation, the 2x2 and single-shift pieces, the
// memory accessors and the wait / signal primitives of a workgroup team, and the Ehrlich-Aberth shift solvers.
// (Round 1's own iteration - single-shift bulges, bulge-major log - lived here; kb_hqr2.hpp replaced it.)
#pragma once
#include "kb_eig.hpp"

namespace kb {


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
#define KB_TEAM_SLOTS 4


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

}  // namespace kb
