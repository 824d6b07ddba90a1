// gfx950 kernels of the bidiagonal divide-and-conquer SVD (algorithm: kb_bdsdc.hpp; replaces the rotation replay of
// the QR-iteration route for scipy.linalg.svd -> zgesdd -> dbdsdc, reference kbdm.py:166).
//
//   k_dc_leaf    one workgroup of two wavefronts per (member, leaf): one-sided Jacobi of the <= 32-row leaf blocks, four
//                lanes per pair of columns
//   k_dc_setup   one workgroup per (member, node) of a depth: deflation, secular equation (one root per thread), Loewner
//                z, the two coefficient matrices CU / CV
//   k_dc_apply   all CUs: the node products  U = Ubasis CU,  V = Vbasis CV  as REAL FP64-MFMA tiles (the bases are block
//                diagonal: a row tile inside one child only contracts over that child's columns)
//   k_dc_final   all CUs: L = Q X and R = P Y (complex x real = real GEMMs on the (2m) x m views of Q and P), columns
//                already in descending order of the singular values
//   k_dc_sv      singular values, Dsqi (kbdm.py:168-186), outputs
// Members of different size have trees of different depth: launch `step` handles depth dc_depth(m) - 1 - step of every
// member that still has one, so every member's root is merged in its own last step.
#pragma once
#include <hip/hip_runtime.h>

#include "kb_bdsdc.hpp"
#include "kbdm_device.h"

// (kb_smem, make_ctx, kb_d4, KB_TU_KC, KB_TU_PITCH come from kbdm_kernels.hpp, which includes this file)

// Real FP64-MFMA tile product: C[i][j] (+)= sum_{k in [k0, k1)} Aop(i, k) Bop(j, k) for a 64 x 64 tile, 256 threads.
// Aop is staged row-fastest (contiguous along i in memory), Bop k-fastest (contiguous along k).  Same LDS layout, operand
// maps and transposed-accumulator trick as mfma_tile_kx; one real MFMA per 16 x 16 x 4 block.
template <class FA, class FB, class FC>
__device__ __forceinline__ void mfma_rtile(FA Aop, FB Bop, FC Cstore, int k0, int k1) {
    __shared__ double s_r[2][2][KB_TU_KC][KB_TU_PITCH];   // [buffer][A|B][k][row]
    const int t = threadIdx.x;
    const int wave = t >> 6, lane = t & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int wr = (wave & 1) * 32, wc = (wave >> 1) * 32;
    const int arow = t & 63, ak = t >> 6;                  // A: rows fastest; (arow, ak) and (arow, ak + 4)
    const int bk = t & 7, brow = t >> 3;                   // B: k fastest;    (brow, bk) and (brow + 32, bk)
    kb_d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (kb_d4){0, 0, 0, 0};
    const int nch = (k1 - k0 + KB_TU_KC - 1) / KB_TU_KC;
    double ga0, ga1, gb0, gb1;
    auto fetch = [&](int ch) {
        const int kb = k0 + ch * KB_TU_KC;
        ga0 = (kb + ak < k1) ? Aop(arow, kb + ak) : 0.0;
        ga1 = (kb + ak + 4 < k1) ? Aop(arow, kb + ak + 4) : 0.0;
        gb0 = (kb + bk < k1) ? Bop(brow, kb + bk) : 0.0;
        gb1 = (kb + bk < k1) ? Bop(brow + 32, kb + bk) : 0.0;
    };
    auto stage = [&](int buf) {
        s_r[buf][0][ak][arow] = ga0; s_r[buf][0][ak + 4][arow] = ga1;
        s_r[buf][1][bk][brow] = gb0; s_r[buf][1][bk][brow + 32] = gb1;
    };
    if (nch > 0) { fetch(0); stage(0); }
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nch) fetch(ch + 1);
#pragma unroll
        for (int ks = 0; ks < KB_TU_KC; ks += 4) {
            double av[2], bv[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                av[a] = s_r[buf][0][ks + lk][wr + a * 16 + li];
                bv[a] = s_r[buf][1][ks + lk][wc + a * 16 + li];
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
                    acc[cb][rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[cb], av[rb], acc[cb][rb], 0, 0, 0);
        }
        if (ch + 1 < nch) stage(buf ^ 1);
        __syncthreads();
    }
    // D'[c][r]: lane (li, lk) holds C(row = wr + rb*16 + li, col = wc + cb*16 + lk + 4 g)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int g = 0; g < 4; ++g) Cstore(wr + rb * 16 + li, wc + cb * 16 + lk + 4 * g, acc[cb][rb][g]);
}

__device__ __forceinline__ DcWs dc_item_ws(const KbItem& it, double* dcarena) { return dc_ws(dcarena + it.dc_off, it.m); }

__global__ void __launch_bounds__(128) k_dc_leaf(const KbItem* __restrict__ items, const int* __restrict__ perm,
                                                 double* varena, double* dcarena, int smem_bytes) {
    const KbItem it = items[perm[blockIdx.y]];
    const int m = it.m, L = dc_depth(m);
    if ((int)blockIdx.x >= (1 << L)) return;
    const DevCtx ctx = make_ctx(smem_bytes);
    const double* d = varena + it.voff + KB_V_D * it.vstride;
    const double* e = varena + it.voff + KB_V_E * it.vstride;
    const DcWs ws = dc_item_ws(it, dcarena);
    const double scale = dc_scale(ctx, d, e, m);
    const DcNode nd = dc_node(m, L, blockIdx.x);
    dc_leaf(ctx, d, e, m, nd, ws.U[L & 1], ws.V[L & 1], ws.D[L & 1], L == 0, scale);
}

__global__ void __launch_bounds__(1024) k_dc_setup(const KbItem* __restrict__ items, const int* __restrict__ perm,
                                                    double* varena, double* dcarena, int step, int* status, int smem_bytes) {
    const int item = perm[blockIdx.y];
    const KbItem it = items[item];
    const int m = it.m, depth = dc_depth(m) - 1 - step;
    if (depth < 0 || (int)blockIdx.x >= (1 << depth)) return;
    const DevCtx ctx = make_ctx(smem_bytes);
    const double* d = varena + it.voff + KB_V_D * it.vstride;
    const double* e = varena + it.voff + KB_V_E * it.vstride;
    const DcWs ws = dc_item_ws(it, dcarena);
    const double scale = dc_scale(ctx, d, e, m);
    const DcNode nd = dc_node(m, depth, blockIdx.x);
    __shared__ int info;
    if (threadIdx.x == 0) info = 0;
    __syncthreads();
    dc_merge_setup(ctx, d, e, ws, nd, (depth + 1) & 1, depth == 0, &info, scale);
    __syncthreads();
    if (threadIdx.x == 0 && info) atomicOr(&status[item], KB_STAT_SVD_NOCONV);
}

// grid (nodes_max * tmax * tmax, 2 sides, members); block 256
__global__ void __launch_bounds__(256) k_dc_apply(const KbItem* __restrict__ items, const int* __restrict__ perm,
                                                   double* dcarena, int step, int tmax) {
    const KbItem it = items[perm[blockIdx.z]];
    const int m = it.m, depth = dc_depth(m) - 1 - step;
    if (depth < 0) return;
    const int idx = blockIdx.x / (tmax * tmax), tt = blockIdx.x % (tmax * tmax);
    if (idx >= (1 << depth)) return;
    const DcNode nd = dc_node(m, depth, idx);
    const int side = blockIdx.y;                       // 0: U (n x n), 1: V (n + sqre)
    const int n = nd.n, nn = side ? n + nd.sqre : n, lo = nd.lo, nl = (n - 1) / 2;
    const int r0 = (tt % tmax) * 64, c0 = (tt / tmax) * 64;
    if (r0 >= nn || c0 >= nn) return;
    const DcWs ws = dc_item_ws(it, dcarena);
    const int src = (depth + 1) & 1;
    const double* Bs = side ? ws.V[src] : ws.U[src];
    const double* Cf = side ? ws.CV : ws.CU;
    double* Out = side ? ws.V[src ^ 1] : ws.U[src ^ 1];
    // block-diagonal basis: rows of child 1 (U: r < nl, V: r <= nl) contract over that child's columns only, rows of
    // child 2 over its own; U's centre row takes the one coefficient row nl
    const int split = side ? nl + 1 : nl;              // first row / column of the second block (U: after the centre)
    int k0 = 0, k1 = nn;
    const int rend = (r0 + 64 < nn) ? r0 + 64 : nn;
    if (rend <= split) k1 = side ? split : split + 1;   // (U: a tile that ends at the centre row may include it)
    else if (r0 >= (side ? split : split + 1)) k0 = side ? split : split + 1;
    mfma_rtile(
        [&](int i, int k) -> double {
            const int r = r0 + i;
            if (r >= nn) return 0.0;
            return side ? dc_vbasis(Bs, m, lo, nl, r, k) : dc_ubasis(Bs, m, lo, nl, r, k);
        },
        [&](int j, int k) -> double {
            const int c = c0 + j;
            return (c < nn) ? Cf[(lo + k) + (size_t)(lo + c) * m] : 0.0;
        },
        [&](int i, int j, double v) {
            const int r = r0 + i, c = c0 + j;
            if (r < nn && c < nn) Out[(lo + r) + (size_t)(lo + c) * m] = v;
        },
        k0, k1);
}

// grid (ceil(2 mmax / 64), ceil(mmax / 64), 2 * members); block 256.  z even: L = Q X (-> A buffer), odd: R = P Y (-> R)
__global__ void __launch_bounds__(256) k_dc_final(const KbItem* __restrict__ items, const int* __restrict__ perm,
                                                   cd* arena, double* dcarena) {
    const KbItem it = items[perm[blockIdx.z >> 1]];
    const int side = blockIdx.z & 1;
    const int m = it.m, m2 = 2 * m;
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    if (r0 >= m2 || c0 >= m) return;
    const DcWs ws = dc_item_ws(it, dcarena);
    const double* Qr = reinterpret_cast<const double*>(arena + it.off[side ? KB_BUF_P : KB_BUF_Q]);
    const double* X = side ? ws.V[0] : ws.U[0];
    double* Out = reinterpret_cast<double*>(arena + it.off[side ? KB_BUF_R : KB_BUF_A]);
    mfma_rtile(
        [&](int i, int k) -> double { return (r0 + i < m2) ? Qr[(r0 + i) + (size_t)k * m2] : 0.0; },
        [&](int j, int k) -> double { return (c0 + j < m) ? X[k + (size_t)(c0 + j) * m] : 0.0; },
        [&](int i, int j, double v) {
            const int r = r0 + i, c = c0 + j;
            if (r < m2 && c < m) Out[r + (size_t)c * m2] = v;
        },
        0, m);
}

__global__ void __launch_bounds__(256) k_dc_sv(const KbItem* __restrict__ items, const int* __restrict__ perm,
                                                double* varena, double* dcarena, double* sv_out) {
    const KbItem it = items[perm[blockIdx.x]];
    const DevCtx ctx = make_ctx(0);
    const int m = it.m;
    double* dv = varena + it.voff;
    const double scale = dc_scale(ctx, dv + KB_V_D * it.vstride, dv + KB_V_E * it.vstride, m);
    const DcWs ws = dc_item_ws(it, dcarena);
    double* s = dv + KB_V_S * it.vstride;
    double* dsqi = dv + KB_V_DSQI * it.vstride;
    // singular values out + the scaling Dsqi = 1/sqrt(s) (q = 0) or 1/sqrt(s + q^2/s)  [kbdm.py:179-186]
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        const double si = ws.D[0][i] * scale;
        s[i] = si;
        if (sv_out) sv_out[it.sv_off + i] = si;
        if (i < it.l) dsqi[i] = (it.q > 0.0) ? 1.0 / sqrt(si + it.q * it.q / si) : 1.0 / sqrt(si);
    }
}
