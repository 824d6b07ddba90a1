// gfx950 kernels of the KBDM ensemble pipeline.  One launch of each kernel covers the whole
// batch of ensemble members ("items"); factorisation kernels give one workgroup to each
// item (largest items first), contraction kernels tile every item over the grid.
//
// Pipeline (reference kbdm.py:64-92 per member; sampling.py:52-70 over members):
//   k_hankel   U^{p-1} from the shared signal                      (kbdm.py:95-130)
//   k_bidiag_panel / k_trail_update / k_svd_fac   blocked Householder bidiagonalisation: panels of
//              32 reflector pairs, FP64-MFMA rank-64 trailing update, unblocked tail   (kbdm.py:166)
//   k_gen      explicit Q, P (and later Qh), columns in registers  (kbdm.py:166,192)
//   k_dc_*     bidiagonal SVD by divide and conquer (kbdm_dc_kernels.hpp), L = Q X, R = P Y,
//              singular values, Dsqi                                (kbdm.py:166-186)
//   k_gemm<1>  T1 = U^p R_        (U^p read straight from the signal: Hankel operand)
//   k_gemm<2>  W  = Dsqi L_^H T1 Dsqi                               (kbdm.py:168-189)
//   k_hess     Hessenberg reduction + Qh + copies                   (kbdm.py:192)
//   k_hqr      eigenvalues (active-block multi-bulge QR)            (kbdm.py:192)
//   k_invit    eigenvectors by inverse iteration                    (kbdm.py:192)
//   k_gemm<3>  G  = Dsqi (Qh X)                                     (kbdm.py:198)
//   k_gemm<4>  B  = R_ G                                            (kbdm.py:198)
//   k_gemm<5>  T  = U0 B          (U0 read from the signal)         (kbdm.py:232)
//   k_epilogue N_k, D_k, A, T2, F, PH, keep mask                    (kbdm.py:71-90, sampling.py:75-97)
#pragma once
#include <hip/hip_runtime.h>

#include "kb_eig.hpp"
#include "kb_hqr_ms.hpp"
#include "kb_hqr2.hpp"
#include "kb_svd.hpp"
#include "kb_panel_team.hpp"
#include "kbdm_device.h"

using namespace kb;

extern __shared__ __attribute__((aligned(16))) char kb_smem[];

__device__ __forceinline__ DevCtx make_ctx(int smem_bytes) {
    DevCtx c;
    c.smem = kb_smem;
    c.smem_bytes = smem_bytes;
    return c;
}

// ------------------------------------------------------------------------------------
// K1: Hankel assembly.  U^{s}[i,j] = c[i+j+s].  Each workgroup stages the 2*TILE-1 signal
// samples its tile needs in LDS (one coalesced read), then streams coalesced 16-byte
// stores.  Hankel matrices are symmetric, so row-major and column-major coincide.
// grid = (tiles_x, tiles_y, B); outputs selected by the launch (pipeline: only U^{p-1}).
constexpr int HK_TILE = 64;

__global__ void __launch_bounds__(256) k_hankel(const KbItem* __restrict__ items, const int* __restrict__ perm,
                                                 const cd* __restrict__ signals, int N, int nout,
                                                 HankelOut o0, HankelOut o1, HankelOut o2) {
    const KbItem it = items[perm ? perm[blockIdx.z] : blockIdx.z];
    const int m = it.m;
    const int r0 = blockIdx.y * HK_TILE, c0 = blockIdx.x * HK_TILE;
    if (r0 >= m || c0 >= m) return;
    __shared__ cd seg[2 * HK_TILE + 8];
    const cd* sig = signals + (size_t)it.sig * N;
    const HankelOut outs[3] = {o0, o1, o2};
    for (int q = 0; q < nout; ++q) {
        const int shift = outs[q].shift;
        cd* dst = outs[q].base + (outs[q].use_item_off ? it.off[outs[q].buf] : it.hk_off);
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * HK_TILE - 1; t += blockDim.x) {
            const int idx = r0 + c0 + shift + t;
            seg[t] = (idx < N) ? sig[idx] : czero();
        }
        __syncthreads();
        // 64x64 tile, thread -> (column-of-memory-row i fastest): element (r, c) at dst[c*m + r]
        for (int e = threadIdx.x; e < HK_TILE * HK_TILE; e += blockDim.x) {
            const int i = e % HK_TILE, j = e / HK_TILE;
            const int r = r0 + i, c = c0 + j;
            if (r < m && c < m) dst[(size_t)c * m + r] = seg[i + j];
        }
    }
}

// ------------------------------------------------------------------------------------
// Householder bidiagonalisation, blocked: for panel p = 0, 1, ... (host loop)
//   k_bidiag_panel_team  a team of T >= 1 workgroups per item: NB reflector pairs, X/Y panels (Q and P buffers are
//                   free at this point), two matrix-vector products with the trailing matrix per column
//   k_trail_update  all CUs: A0[NB:, NB:] -= [V | X] [Y | U]^H  with FP64 MFMA 16x16x4 tiles
// then k_svd_fac finishes the last (< NB + NX) columns unblocked.
// Left vectors stay in A, right vectors go to the R buffer (free until the sort).
// The panel for a team of T workgroups per member (kb_team.hpp, kb_panel_team.hpp; T = 1: one workgroup, no waiting).
// grid = 8 * ceil(count / 8) * T: block b belongs to slot b & 7 (blocks of equal slot share an XCD when the dispatcher deals
// them round robin - faster, never required), within the slot consecutive groups of T blocks are the teams.  `pos0` is the
// chunk's first position in the sorted order (the control words are indexed by position), `epoch0` the synchronisations
// a team has behind it when this launch starts (every panel of a stage makes the same number).
__device__ __forceinline__ bool team_geom(int T, int count, int& team, int& role) {
    const int b = blockIdx.x;
    const int idx = b >> 3;
    role = idx % T;
    team = (idx / T) * 8 + (b & 7);
    return team < count;
}
constexpr int KB_BIDIAG_TEAM_SYNCS = 2 * KB_NB + 1;     // per panel
constexpr int KB_HESS_TEAM_SYNCS = KB_NB + 1;

__global__ void __launch_bounds__(1024) k_bidiag_panel_team(const KbItem* __restrict__ items, const int* __restrict__ perm,
                                                             cd* arena, double* varena, int panel, int smem_bytes, int T,
                                                             int count, PanelTeamCtl* ctl, int zr, int* status) {
    int team, role;
    if (!team_geom(T, count, team, role)) return;
    const int item = perm[team];
    const KbItem it = items[item];
    const int m = it.m;
    if (panel >= bidiag_num_panels(m)) return;
    const DevCtx ctx = make_ctx(smem_bytes);
    PanelTeam<DevCtx> tm;
    tm.T = T; tm.role = role; tm.ctl = ctl + team; tm.failed = 0;
    tm.epoch = (unsigned)(panel * KB_BIDIAG_TEAM_SYNCS);
    tm.xb = arena + it.off[KB_BUF_H];                      // free until the unitary factors are accumulated
    if (T > 1 && __hip_atomic_load(&tm.ctl->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    const int p0 = panel * KB_NB;
    cd* A = arena + it.off[KB_BUF_A] + p0 + (size_t)p0 * m;
    cd* UR = arena + it.off[KB_BUF_R] + p0 + (size_t)p0 * m;
    cd* X = arena + it.off[KB_BUF_Q];
    cd* Y = arena + it.off[KB_BUF_P];
    double* dv = varena + it.voff;
    double* d = dv + KB_V_D * it.vstride + p0;
    double* e = dv + KB_V_E * it.vstride + p0;
    cd* tauq = reinterpret_cast<cd*>(dv + KB_V_TAUQ * it.vstride) + p0;
    cd* taup = reinterpret_cast<cd*>(dv + KB_V_TAUP * it.vstride) + p0;
    if (!bidiag_panel_team(ctx, tm, m - p0, A, m, d, e, tauq, taup, UR, m, X, Y, m, zr) && threadIdx.x == 0)
        atomicOr(&status[item], KB_STAT_SVD_NOCONV);
}

// Rank-64 update of one 64 x 64 tile of C on FP64 MFMA:  C[r, c] -= sum_k Aop(r, k) conj(Bop(c, k)), k < 2 NB.
// The operands go through LDS: the workgroup stages them in chunks of 8 k (re and im planes, [k][row], row
// pitch 80 doubles so that the four k-groups of a wavefront's ds_read_b64 hit disjoint banks), double
// buffered with the next chunk's global loads in flight under the current chunk's MFMAs; every operand element
// is fetched from L2 once per workgroup instead of once per wavefront.  Each of the four wavefronts owns a
// 32 x 32 sub-block as 2 x 2 v_mfma_f64_16x16x4_f64 tiles (real and imaginary accumulators).  The MFMA computes
// D'[c][r] (output column on the MFMA row index) so that the 16 lanes of a quarter-wave hold 16 consecutive
// ROWS of C: coalesced 256-byte read-modify-write.
//   f64 MFMA operand maps (cdna_hip_programming.md 3): A: lane l -> A[l & 15][l >> 4],
//   B: lane l -> B[l >> 4][l & 15],  C/D: col = l & 15, row = (l >> 4) + 4 * reg.
//   Aop(r, k), Bop(c, k): accessors with r, c relative to the tile origin (they return 0 outside the matrix).
typedef double kb_d4 __attribute__((ext_vector_type(4)));
constexpr int KB_WY_MIN = 192;    // members at least this large get their unitary factors from the blocked (k_wy_*) path
constexpr int KB_TU_KC = 8;       // k per staged chunk
constexpr int KB_TU_PITCH = 80;   // doubles per k row in LDS
constexpr int KB_TU_PF = 1;       // chunks of operand loads in flight ahead of the MFMAs (2 and 3 measured: 152-165 VGPRs, 1 % slower in flight)

// `nch` = number of 8-deep k chunks (a runtime value: the blocked reductions use 2 NB / 8, the compact-WY generation of
// the unitary factors uses the panel height / 8 and NB / 8); STORE = false: C -= product, true: C = product.
// KFAST: the operands are read k-fastest (thread t stages k = t & 7 of rows t >> 3 and (t >> 3) + 32): for products whose
// operands are contiguous along k in memory (A^H B: both factors column-major with k = the row index).
typedef double kb_tu_stage[2][2][KB_TU_KC][KB_TU_PITCH];       // [A|B][re|im][k][row]; a tile product needs two (double buffer)
template <bool STORE, bool KFA, bool KFB, class FA, class FB, class FC>
__device__ __forceinline__ void mfma_tile_ks2(kb_tu_stage* s_op, FA Aop, FB Bop, FC Cptr, int NCH) {
    const int t = threadIdx.x;
    const int wave = t >> 6, lane = t & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int wr = (wave & 1) * 32, wc = (wave >> 1) * 32;
    // staging: thread t holds (row srow, k = sk) and (row srow + srow2, k = sk + sk2) of a chunk, per operand
    const int srowa = KFA ? (t >> 3) : (t & 63), ska = KFA ? (t & 7) : (t >> 6);
    constexpr int srowa2 = KFA ? 32 : 0, ska2 = KFA ? 0 : 4;
    const int srowb = KFB ? (t >> 3) : (t & 63), skb = KFB ? (t & 7) : (t >> 6);
    constexpr int srowb2 = KFB ? 32 : 0, skb2 = KFB ? 0 : 4;
    kb_d4 acc_re[2][2], acc_im[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) { acc_re[a][b] = (kb_d4){0, 0, 0, 0}; acc_im[a][b] = (kb_d4){0, 0, 0, 0}; }
    // Operand loads run KB_TU_PF chunks ahead of the MFMAs (registers: PF sets of two elements per operand and thread; a
    // set is free again once its chunk sits in LDS).  One chunk of MFMAs lasts 0.85 us; with four ensembles in flight a
    // load takes about 3 us, and one chunk of cover left the tile waiting for its operands 70 % of the time.
    constexpr int PF = KB_TU_PF;
    cd ga[PF][2], gb[PF][2];
#pragma unroll
    for (int u = 0; u < PF; ++u)
        if (u < NCH) {
            const int kna = u * KB_TU_KC + ska, knb = u * KB_TU_KC + skb;
            ga[u][0] = Aop(srowa, kna); ga[u][1] = Aop(srowa + srowa2, kna + ska2);
            gb[u][0] = Bop(srowb, knb); gb[u][1] = Bop(srowb + srowb2, knb + skb2);
        }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        s_op[0][0][0][ska + ska2 * j][srowa + srowa2 * j] = ga[0][j].x; s_op[0][0][1][ska + ska2 * j][srowa + srowa2 * j] = ga[0][j].y;
        s_op[0][1][0][skb + skb2 * j][srowb + srowb2 * j] = gb[0][j].x; s_op[0][1][1][skb + skb2 * j][srowb + srowb2 * j] = gb[0][j].y;
    }
    __syncthreads();
    for (int ch0 = 0; ch0 < NCH; ch0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int ch = ch0 + u;
            if (ch < NCH) {                                  // (uniform)
                const int buf = ch & 1;
                if (ch + PF < NCH) {                         // set u held chunk ch, staged one step ago: reuse it for chunk ch + PF
                    const int kna = (ch + PF) * KB_TU_KC + ska, knb = (ch + PF) * KB_TU_KC + skb;
                    ga[u][0] = Aop(srowa, kna); ga[u][1] = Aop(srowa + srowa2, kna + ska2);
                    gb[u][0] = Bop(srowb, knb); gb[u][1] = Bop(srowb + srowb2, knb + skb2);
                }
#pragma unroll
                for (int ks = 0; ks < KB_TU_KC; ks += 4) {
                    double ar[2], ai[2], br[2], bi[2];
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        ar[a] = s_op[buf][0][0][ks + lk][wr + a * 16 + li];
                        ai[a] = s_op[buf][0][1][ks + lk][wr + a * 16 + li];
                        br[a] = s_op[buf][1][0][ks + lk][wc + a * 16 + li];
                        bi[a] = s_op[buf][1][1][ks + lk][wc + a * 16 + li];
                    }
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb) {
                            // D'[c][r] += Bop(c,k) (MFMA A operand) * Aop(r,k) (MFMA B operand), complex with conj(Bop)
                            acc_re[cb][rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(br[cb], ar[rb], acc_re[cb][rb], 0, 0, 0);
                            acc_re[cb][rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(bi[cb], ai[rb], acc_re[cb][rb], 0, 0, 0);
                            acc_im[cb][rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(br[cb], ai[rb], acc_im[cb][rb], 0, 0, 0);
                            acc_im[cb][rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(-bi[cb], ar[rb], acc_im[cb][rb], 0, 0, 0);
                        }
                }
                if (ch + 1 < NCH) {
                    const int un = (u + 1) % PF;             // the set that holds chunk ch + 1 (a constant once the loop is unrolled)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        s_op[buf ^ 1][0][0][ska + ska2 * j][srowa + srowa2 * j] = ga[un][j].x; s_op[buf ^ 1][0][1][ska + ska2 * j][srowa + srowa2 * j] = ga[un][j].y;
                        s_op[buf ^ 1][1][0][skb + skb2 * j][srowb + srowb2 * j] = gb[un][j].x; s_op[buf ^ 1][1][1][skb + skb2 * j][srowb + srowb2 * j] = gb[un][j].y;
                    }
                }
                __syncthreads();
            }
        }
    }
    // D' element (MFMA row = C column offset, MFMA col = C row offset): col = li, row = lk + 4*reg
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                cd* pc = Cptr(wr + rb * 16 + li, wc + cb * 16 + lk + 4 * g);
                if (pc) {
                    if (STORE) {
                        *pc = mk(acc_re[cb][rb][g], acc_im[cb][rb][g]);
                    } else {
                        cd v = *pc;
                        v.x -= acc_re[cb][rb][g];
                        v.y -= acc_im[cb][rb][g];
                        *pc = v;
                    }
                }
            }
}

template <bool STORE, bool KFAST, class FA, class FB, class FC>
__device__ __forceinline__ void mfma_tile_ks(kb_tu_stage* s_op, FA Aop, FB Bop, FC Cptr, int NCH) {
    mfma_tile_ks2<STORE, KFAST, KFAST>(s_op, Aop, Bop, Cptr, NCH);
}
template <bool STORE, bool KFAST, class FA, class FB, class FC>
__device__ __forceinline__ void mfma_tile_kx(FA Aop, FB Bop, FC Cptr, int NCH) {
    __shared__ kb_tu_stage s_op[2];
    mfma_tile_ks<STORE, KFAST>(s_op, Aop, Bop, Cptr, NCH);
}
template <bool STORE, class FA, class FB, class FC>
__device__ __forceinline__ void mfma_tile_k(FA Aop, FB Bop, FC Cptr, int NCH) {
    mfma_tile_kx<STORE, false>(Aop, Bop, Cptr, NCH);
}
template <class FA, class FB, class FC>
__device__ __forceinline__ void mfma_rank2nb_tile(FA Aop, FB Bop, FC Cptr) {
    mfma_tile_k<false>(Aop, Bop, Cptr, 2 * KB_NB / KB_TU_KC);
}

// Trailing update of one panel:  C[r, c] -= sum_k Aop[r, k] conj(Bop[c, k]),  r, c in [NB, n),
// Aop = [V | X], Bop = [Y | U], K = 2 NB.  One workgroup = 64 x 64 block of C (mfma_rank2nb_tile).
__global__ void __launch_bounds__(256) k_trail_update(const KbItem* __restrict__ items, const int* __restrict__ perm,
                                                       cd* arena, int panel) {
    const KbItem it = items[perm[blockIdx.z]];
    const int m = it.m;
    if (panel >= bidiag_num_panels(m)) return;
    const int p0 = panel * KB_NB;
    const int n = m - p0;                        // trailing block size at panel start
    const int nn = n - KB_NB;                    // updated block is nn x nn at local offset NB
    if (blockIdx.x * 64 >= nn || blockIdx.y * 64 >= nn) return;
    const int r0 = KB_NB + blockIdx.x * 64;      // tile origin, local to the trailing block
    const int c0 = KB_NB + blockIdx.y * 64;
    const cd* A = arena + it.off[KB_BUF_A] + p0 + (size_t)p0 * m;    // V in columns 0..NB-1
    const cd* UR = arena + it.off[KB_BUF_R] + p0 + (size_t)p0 * m;   // U
    const cd* X = arena + it.off[KB_BUF_Q];
    const cd* Y = arena + it.off[KB_BUF_P];
    cd* C = arena + it.off[KB_BUF_A] + p0 + (size_t)p0 * m;
    mfma_rank2nb_tile(
        [&](int i, int k) -> cd {
            const int r = r0 + i;
            if (r >= n) return czero();
            return (k >= KB_NB) ? X[r + (size_t)(k - KB_NB) * m] : A[r + (size_t)k * m];
        },
        [&](int i, int k) -> cd {
            const int c = c0 + i;
            if (c >= n) return czero();
            return (k >= KB_NB) ? UR[c + (size_t)(k - KB_NB) * m] : Y[c + (size_t)k * m];
        },
        [&](int i, int jx) -> cd* {
            const int r = r0 + i, c = c0 + jx;
            return (r < n && c < n) ? &C[r + (size_t)c * m] : nullptr;
        });
}

// One workgroup per item: the remaining columns, unblocked.
__global__ void __launch_bounds__(1024) k_svd_fac(const KbItem* __restrict__ items, const int* __restrict__ perm,
                                                   cd* arena, double* varena, int smem_bytes, int blocked) {
    const KbItem it = items[perm[blockIdx.x]];
    const DevCtx ctx = make_ctx(smem_bytes);
    const int m = it.m;
    cd* A = arena + it.off[KB_BUF_A];
    cd* UR = arena + it.off[KB_BUF_R];
    double* dv = varena + it.voff;
    double* d = dv + KB_V_D * it.vstride;
    double* e = dv + KB_V_E * it.vstride;
    cd* tauq = reinterpret_cast<cd*>(dv + KB_V_TAUQ * it.vstride);
    cd* taup = reinterpret_cast<cd*>(dv + KB_V_TAUP * it.vstride);
    const int k0 = blocked ? bidiag_num_panels(m) * KB_NB : 0;
    // behind the panels fewer than NB + NX columns are left: that block is reduced in LDS (the same arithmetic)
    if (k0 > 0 && bidiag_tail_lds_bytes(m - k0, blockDim.x >> 6, 64) <= ctx.scratch_bytes())
        bidiag_tail_lds(ctx, m, A, m, d, e, tauq, taup, UR, m, k0);
    else
        bidiag(ctx, m, A, m, d, e, tauq, taup, UR, m, k0);
}

// Explicit unitary factors, columns spread over gridDim.x workgroups per item and matrix.
//   mode 0: blockIdx.z = 0 -> Q (from A, tauq) into Q;  blockIdx.z = 1 -> P (from R buffer, taup) into P
//   mode 1: Qh (from the Hessenberg vectors in P, tauh) into Q
template <int MAXC>
__global__ void __launch_bounds__(256) k_gen(const KbItem* __restrict__ items, const int* __restrict__ perm,
                                              cd* arena, double* varena, int mode, int wy) {
    const KbItem it = items[perm[blockIdx.y]];
    const DevCtx ctx = make_ctx(0);
    double* dv = varena + it.voff;
    int n, nref, shift;
    const cd* V;
    const cd* tau;
    cd* Out;
    if (mode == 0 && blockIdx.z == 0) {
        n = it.m; nref = n; shift = 0;
        V = arena + it.off[KB_BUF_A]; tau = reinterpret_cast<const cd*>(dv + KB_V_TAUQ * it.vstride);
        Out = arena + it.off[KB_BUF_Q];
    } else if (mode == 0) {
        n = it.m; nref = n - 1; shift = 1;
        V = arena + it.off[KB_BUF_R]; tau = reinterpret_cast<const cd*>(dv + KB_V_TAUP * it.vstride);
        Out = arena + it.off[KB_BUF_P];
    } else {
        n = it.l; nref = n - 2; shift = 1;
        V = arena + it.off[KB_BUF_P]; tau = reinterpret_cast<const cd*>(dv + KB_V_TAUQ * it.vstride);
        Out = arena + it.off[KB_BUF_Q];
    }
    if (n > MAXC * 64) return;                       // host picks MAXC from the largest item
    if (wy && n >= KB_WY_MIN) return;                // large members: blocked accumulation (k_wy_*)
    const int cpw = (n + gridDim.x - 1) / gridDim.x;
    const int c0 = blockIdx.x * cpw;
    const int c1 = (c0 + cpw < n) ? c0 + cpw : n;
    if (c0 >= n) return;
    gen_unitary_cols<DevCtx, MAXC>(ctx, n, nref, shift, V, n, tau, Out, n, c0, c1);
}

// ------------------------------------------------------------------------------------
// Explicit unitary factors by BLOCKED backward accumulation (the zungqr / zungbr organisation) on FP64 MFMA, for
// members with n >= KB_WY_MIN (smaller ones keep the per-column kernel k_gen).  Q = H_0 H_1 ... H_{nref-1} is built
// from the last block of KB_WYB = 64 reflectors to the first: with the block reflector  H_kb .. H_kb+63 = I - V T V^H
// (compact WY, T upper triangular),  Out[p0:, p0:] <- (I - V T V^H) Out[p0:, p0:],  p0 = kb + shift.
//   k_wy_tfac   ONE launch for all blocks, members and matrices: G = V^H V of the block as an MFMA tile (k = panel
//               height), then T from G and tau (zlarft, forward / columnwise) in LDS; T to the workspace;
//   k_wy_apply  ONE launch per block: a workgroup owns 64 columns of the trailing matrix C = Out[p0:, p0:]:
//               Z = V^H C (MFMA tile, k = panel height), Y = T Z (64 x 64 x 64), both kept in LDS, then
//               C -= V Y row tile by row tile (rank-64 MFMA updates).  Every column tile is independent of the others,
//               so a block costs one launch and the trailing matrix is read twice (the second time out of the L2).
// The result is the same product of reflectors as k_gen's (rounding differs); flops 32/3 n^3 (Q, P) + 16/3 l^3 (Qh).
constexpr int KB_WYB = 64;
struct WyGeom {
    int n, nref, shift;
    const cd* V;          // reflector k: column k, rows > k + shift (1 at row k + shift, 0 above)
    const cd* tau;
    cd* Out;              // n x n
    cd* Tws;              // workspace of this matrix: T of every block (64 x 64 each, T[j][k] at j * 64 + k)
};
__device__ __forceinline__ bool wy_geom(const KbItem& it, cd* arena, double* varena, int mode, int z, WyGeom& g) {
    double* dv = varena + it.voff;
    cd* wsbase;
    if (mode == 0) {
        g.n = it.m;
        if (z == 0) { g.nref = g.n; g.shift = 0; g.V = arena + it.off[KB_BUF_A]; g.tau = reinterpret_cast<const cd*>(dv + KB_V_TAUQ * it.vstride); g.Out = arena + it.off[KB_BUF_Q]; }
        else { g.nref = g.n - 1; g.shift = 1; g.V = arena + it.off[KB_BUF_R]; g.tau = reinterpret_cast<const cd*>(dv + KB_V_TAUP * it.vstride); g.Out = arena + it.off[KB_BUF_P]; }
        wsbase = arena + it.off[KB_BUF_H];            // free until the Hessenberg reduction
    } else {
        g.n = it.l; g.nref = g.n - 2; g.shift = 1;
        g.V = arena + it.off[KB_BUF_P]; g.tau = reinterpret_cast<const cd*>(dv + KB_V_TAUQ * it.vstride); g.Out = arena + it.off[KB_BUF_Q];
        wsbase = arena + it.off[KB_BUF_A];            // L is dead after k_gemm<2>, B is written by k_gemm<4>
    }
    // 2 ceil(n / 64) 4096 <= n^2 elements for every n >= KB_WY_MIN = 192
    g.Tws = wsbase + (size_t)z * ((g.n + KB_WYB - 1) / KB_WYB) * (KB_WYB * KB_WYB);
    return g.n >= KB_WY_MIN;
}
__device__ __forceinline__ cd wy_v(const WyGeom& g, int r, int k) {          // element (r, k) of the reflector matrix
    if (k >= g.nref || r >= g.n) return czero();
    const int p = k + g.shift;
    return r > p ? g.V[r + (size_t)k * g.n] : (r == p ? mk(1.0, 0.0) : czero());
}
// block handled at step s (s = 0: the last block); false if this member has fewer blocks
__device__ __forceinline__ bool wy_block(const WyGeom& g, int step, int& blk, int& kb, int& p0) {
    const int nblk = (g.nref + KB_WYB - 1) / KB_WYB;
    if (step >= nblk) return false;
    blk = nblk - 1 - step;
    kb = blk * KB_WYB;
    p0 = kb + g.shift;
    return true;
}

__global__ void __launch_bounds__(256) k_wy_init(const KbItem* __restrict__ items, const int* __restrict__ perm, cd* arena,
                                                  double* varena, int mode) {
    const KbItem it = items[perm[blockIdx.y]];
    WyGeom g;
    if (!wy_geom(it, arena, varena, mode, blockIdx.z, g)) return;
    const size_t tot = (size_t)g.n * g.n;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(e % g.n), c = (int)(e / g.n);
        g.Out[e] = (r == c) ? mk(1.0, 0.0) : czero();
    }
}

constexpr int KB_WY_LDS = (int)(2 * sizeof(kb_tu_stage) + KB_WYB * (KB_WYB + 1) * sizeof(cd) + KB_WYB * sizeof(cd));
// k_wy_apply: Z / Y without padding, row i rotated by i instead (the same bank spread): 84 KB - a tile of k_ab_iter (75 KB)
// fits beside it on the CU
constexpr int KB_WY_APPLY_LDS = (int)(2 * sizeof(kb_tu_stage) + KB_WYB * KB_WYB * sizeof(cd));

__global__ void __launch_bounds__(256) k_wy_tfac(const KbItem* __restrict__ items, const int* __restrict__ perm, cd* arena,
                                                  double* varena, int mode) {
    const KbItem it = items[perm[blockIdx.y]];
    WyGeom g;
    if (!wy_geom(it, arena, varena, mode, blockIdx.z, g)) return;
    const int nblk = (g.nref + KB_WYB - 1) / KB_WYB;
    const int blk = blockIdx.x;
    if (blk >= nblk) return;
    const int kb = blk * KB_WYB, p0 = kb + g.shift, nrows = g.n - p0;
    extern __shared__ __align__(16) unsigned char wy_smem[];
    kb_tu_stage* s_op = reinterpret_cast<kb_tu_stage*>(wy_smem);
    cd (*G)[KB_WYB + 1] = reinterpret_cast<cd (*)[KB_WYB + 1]>(wy_smem + 2 * sizeof(kb_tu_stage));
    cd* col = reinterpret_cast<cd*>(wy_smem + 2 * sizeof(kb_tu_stage) + KB_WYB * (KB_WYB + 1) * sizeof(cd));
    // G[i][l] = sum_r V[r][i] conj(V[r][l]) = (V^H v_i)[l]; both operands contiguous along the contraction index
    mfma_tile_ks<true, true>(
        s_op,
        [&](int i, int kk) -> cd { return kk < nrows ? wy_v(g, p0 + kk, kb + i) : czero(); },
        [&](int j, int kk) -> cd { return kk < nrows ? wy_v(g, p0 + kk, kb + j) : czero(); },
        [&](int i, int j) -> cd* { return &G[i][j]; },
        (nrows + KB_TU_KC - 1) / KB_TU_KC);
    __syncthreads();
    // zlarft, forward / columnwise:  T(0:i, i) = -tau_i T(0:i, 0:i) (V^H v_i)(0:i),  T(i, i) = tau_i.  In place: column i of T
    // replaces row i of G (G[i][t] <- T[t][i], t <= i); a step reads only row i of G and finished columns of T.
    // (four threads per row of the column: a step is a dot product of at most 16 terms per thread + two shuffles)
    const int t = threadIdx.x, row = t >> 2, part = t & 3;
    for (int i = 0; i < KB_WYB; ++i) {
        const cd ti = (kb + i < g.nref) ? g.tau[kb + i] : czero();
        if (t < i) col[t] = -(ti * G[i][t]);
        __syncthreads();
        cd acc = czero();
        if (row < i)
            for (int l = row + part; l < i; l += 4) cfma(acc, G[l][row], col[l]);
        acc.x += __shfl_xor(acc.x, 1, 64); acc.y += __shfl_xor(acc.y, 1, 64);
        acc.x += __shfl_xor(acc.x, 2, 64); acc.y += __shfl_xor(acc.y, 2, 64);
        if (row < i && part == 0) G[i][row] = acc;      // (row i of G was last read before the barrier above)
        if (t == i) G[i][i] = ti;
        __syncthreads();
    }
    cd* Tg = g.Tws + (size_t)blk * (KB_WYB * KB_WYB);
    for (int e = t; e < KB_WYB * KB_WYB; e += blockDim.x) {
        const int j = e / KB_WYB, k = e % KB_WYB;
        Tg[e] = (k >= j) ? G[k][j] : czero();
    }
}

__global__ void __launch_bounds__(256) k_wy_apply(const KbItem* __restrict__ items, const int* __restrict__ perm, cd* arena,
                                                   double* varena, int mode, int step) {
    const KbItem it = items[perm[blockIdx.y]];
    WyGeom g;
    if (!wy_geom(it, arena, varena, mode, blockIdx.z, g)) return;
    int blk, kb, p0;
    if (!wy_block(g, step, blk, kb, p0)) return;
    const int nrows = g.n - p0, ncols = g.n - p0;
    const int c0 = blockIdx.x * 64;
    if (c0 >= ncols) return;
    extern __shared__ __align__(16) unsigned char wy_smem[];
    kb_tu_stage* s_op = reinterpret_cast<kb_tu_stage*>(wy_smem);
    cd (*Zr)[KB_WYB] = reinterpret_cast<cd (*)[KB_WYB]>(wy_smem + 2 * sizeof(kb_tu_stage));     // [column][reflector + column mod 64]
#define Zs_(i_, j_) Zr[(i_)][((j_) + (i_)) & (KB_WYB - 1)]
    const cd* Tg = g.Tws + (size_t)blk * (KB_WYB * KB_WYB);
    // Zs[c][j] = sum_r C[r][c] conj(V[r][j]) = (V^H C)[j][c]
    mfma_tile_ks<true, true>(
        s_op,
        [&](int i, int kk) -> cd { return (kk < nrows && c0 + i < ncols) ? g.Out[(p0 + kk) + (size_t)(p0 + c0 + i) * g.n] : czero(); },
        [&](int j, int kk) -> cd { return kk < nrows ? wy_v(g, p0 + kk, kb + j) : czero(); },
        [&](int i, int j) -> cd* { return &Zs_(i, j); },
        (nrows + KB_TU_KC - 1) / KB_TU_KC);
    __syncthreads();
    // Yc[c][j] = conj( sum_k T[j][k] Z[k][c] ) = sum_k conj(Zs[c][k]) conj(T[j][k]); written over Zs (the product's operand
    // reads end at the barrier that closes its last chunk)
    mfma_tile_ks<true, false>(
        s_op,
        [&](int i, int k) -> cd { return conj(Zs_(i, k)); },
        [&](int j, int k) -> cd { return Tg[j * KB_WYB + k]; },
        [&](int i, int j) -> cd* { return &Zs_(i, j); },
        KB_WYB / KB_TU_KC);
    __syncthreads();
    // C[r][c] -= sum_j V[r][j] Y[j][c],  Y[j][c] = conj(Yc[c][j])
    for (int r0 = 0; r0 < nrows; r0 += 64)
        mfma_tile_ks<false, false>(
            s_op,
            [&](int i, int j) -> cd { return wy_v(g, p0 + r0 + i, kb + j); },
            [&](int i, int j) -> cd { return Zs_(i, j); },
            [&](int i, int jx) -> cd* {
                const int r = r0 + i, c = c0 + jx;
                return (r < nrows && c < ncols) ? &g.Out[(p0 + r) + (size_t)(p0 + c) * g.n] : nullptr;
            },
            KB_WYB / KB_TU_KC);
}
#undef Zs_

#include "kbdm_dc_kernels.hpp"

// ------------------------------------------------------------------------------------
// Batched complex GEMM, 32x32 tile per workgroup (v0: FP64 vector FMAs through LDS tiles).
//   AMODE 0: A is M x K column-major          AMODE 1: A_op = A^H, A stored K x M
//   AMODE 2: A_op[i,k] = signal[i + k + shift] (Hankel operand generated on the fly)
struct GemmArgs {
    int M, N, K;
    const cd* A; int lda; int amode; int shift;
    const cd* B; int ldb;
    cd* C; int ldc;
    const double* rs; const double* cs;   // optional row / column scalings of C
};

template <int STAGE>
__device__ __forceinline__ GemmArgs gemm_setup(const KbItem& it, const cd* signals, int N, int p,
                                               cd* arena, double* varena) {
    GemmArgs g;
    const int m = it.m, l = it.l;
    const double* dsqi = varena + it.voff + KB_V_DSQI * it.vstride;
    const cd* sig = signals + (size_t)it.sig * N;
    g.rs = nullptr; g.cs = nullptr; g.shift = 0;
    if (STAGE == 1) {        // T1 = U^p R_
        g.M = m; g.N = l; g.K = m; g.A = sig; g.lda = 0; g.amode = 2; g.shift = p;
        g.B = arena + it.off[KB_BUF_R]; g.ldb = m; g.C = arena + it.off[KB_BUF_Q]; g.ldc = m;
    } else if (STAGE == 2) { // W = Dsqi L_^H T1 Dsqi
        g.M = l; g.N = l; g.K = m; g.A = arena + it.off[KB_BUF_A]; g.lda = m; g.amode = 1;
        g.B = arena + it.off[KB_BUF_Q]; g.ldb = m; g.C = arena + it.off[KB_BUF_P]; g.ldc = l;
        g.rs = dsqi; g.cs = dsqi;
    } else if (STAGE == 3) { // G = Dsqi (Qh X)
        g.M = l; g.N = l; g.K = l; g.A = arena + it.off[KB_BUF_Q]; g.lda = l; g.amode = 0;
        g.B = arena + it.off[KB_BUF_H]; g.ldb = l; g.C = arena + it.off[KB_BUF_P]; g.ldc = l;
        g.rs = dsqi;
    } else if (STAGE == 4) { // B = R_ G
        g.M = m; g.N = l; g.K = l; g.A = arena + it.off[KB_BUF_R]; g.lda = m; g.amode = 0;
        g.B = arena + it.off[KB_BUF_P]; g.ldb = l; g.C = arena + it.off[KB_BUF_A]; g.ldc = m;
    } else {                 // T = U0 B
        g.M = m; g.N = l; g.K = m; g.A = sig; g.lda = 0; g.amode = 2; g.shift = 0;
        g.B = arena + it.off[KB_BUF_A]; g.ldb = m; g.C = arena + it.off[KB_BUF_Q]; g.ldc = m;
    }
    return g;
}

constexpr int GT = 64;   // tile edge

// One 64 x 64 tile of C per workgroup on FP64 MFMA (mfma_tile_ks2: operands staged through LDS in chunks of 8 k, each
// operand in the order that is contiguous in memory); the row / column scalings go into the operands while they are
// staged (Dsqi is applied to every term instead of to the sum: the same product up to rounding).
template <int STAGE>
__global__ void __launch_bounds__(256) k_gemm(const KbItem* __restrict__ items, const int* __restrict__ perm,
                                               const cd* __restrict__ signals, int N, int p, cd* arena,
                                               double* varena) {
    const KbItem it = items[perm[blockIdx.z]];
    const GemmArgs g = gemm_setup<STAGE>(it, signals, N, p, arena, varena);
    const int i0 = blockIdx.x * GT, j0 = blockIdx.y * GT;
    if (i0 >= g.M || j0 >= g.N) return;
    __shared__ kb_tu_stage s_op[2];
    constexpr int AMODE = (STAGE == 1 || STAGE == 5) ? 2 : (STAGE == 2 ? 1 : 0);
    constexpr bool KFA = AMODE != 0;                 // A^H and the Hankel operand are contiguous along k, a plain A along rows
    const int nch = (g.K + KB_TU_KC - 1) / KB_TU_KC;
    mfma_tile_ks2<true, KFA, true>(
        s_op,
        [&](int i, int k) -> cd {
            const int gi = i0 + i;
            if (gi >= g.M || k >= g.K) return czero();
            cd a;
            if (AMODE == 0) a = g.A[gi + (size_t)k * g.lda];
            else if (AMODE == 1) a = conj(g.A[k + (size_t)gi * g.lda]);
            else a = g.A[gi + k + g.shift];
            return g.rs ? g.rs[gi] * a : a;
        },
        [&](int j, int k) -> cd {                      // the tile computes sum_k Aop(i, k) conj(Bop(j, k))
            const int gj = j0 + j;
            if (gj >= g.N || k >= g.K) return czero();
            const cd b = conj(g.B[k + (size_t)gj * g.ldb]);
            return g.cs ? g.cs[gj] * b : b;
        },
        [&](int i, int j) -> cd* {
            const int gi = i0 + i, gj = j0 + j;
            return (gi < g.M && gj < g.N) ? &g.C[gi + (size_t)gj * g.ldc] : nullptr;
        },
        nch);
}


// ------------------------------------------------------------------------------------
// Blocked Hessenberg reduction of W (KB_BUF_P): for panel p = 0, 1, ... (host loop)
//   k_hess_panel_team   a team of T >= 1 workgroups per item: NB reflectors, Y -> KB_BUF_Q; VT, MT -> KB_BUF_H (one pass over the
//                  rows of A0 below the panel's first row per column)
//                  ... and, in the same launch, the rows above the panel: their Y = A0 VT and their entries of the panel's columns
//   k_hess_z       all CUs: Z = A0^H VT - V MT for the columns right of the panel -> KB_BUF_H
//   k_hess_update  all CUs: W[:, p0+NB:] -= [Y | V] [V | Z]^H  (FP64 MFMA tiles)
// then k_hess finishes unblocked, extracts the work copy (KB_BUF_H) and ||H||_inf.
__global__ void __launch_bounds__(1024) k_hess_panel_team(const KbItem* __restrict__ items, const int* __restrict__ perm,
                                                           cd* arena, double* varena, int panel, int smem_bytes, int T,
                                                           int count, PanelTeamCtl* ctl, int zr, int* status) {
    int team, role;
    if (!team_geom(T, count, team, role)) return;
    const int item = perm[team];
    const KbItem it = items[item];
    const int n = it.l;
    if (panel >= bidiag_num_panels(n)) return;
    const DevCtx ctx = make_ctx(smem_bytes);
    PanelTeam<DevCtx> tm;
    tm.T = T; tm.role = role; tm.ctl = ctl + team; tm.failed = 0;
    tm.epoch = (unsigned)(panel * KB_HESS_TEAM_SYNCS);
    tm.xb = arena + it.off[KB_BUF_A];                      // L is dead after k_gemm<2>, B is written by k_gemm<4>
    if (T > 1 && __hip_atomic_load(&tm.ctl->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    const int p0 = panel * KB_NB;
    cd* W = arena + it.off[KB_BUF_P];
    cd* Y = arena + it.off[KB_BUF_Q];
    cd* Z = arena + it.off[KB_BUF_H];          // Z (n x NB), then MT (NB x NB), then VT (n x NB): n >= NB + NX
    cd* tauh = reinterpret_cast<cd*>(varena + it.voff + KB_V_TAUQ * it.vstride) + p0;
    if (!hess_panel_team(ctx, tm, n, W, n, p0, tauh, Y, n, Z + (size_t)n * KB_NB + KB_NB * KB_NB, n, Z + (size_t)n * KB_NB, zr) &&
        threadIdx.x == 0)
        atomicOr(&status[item], KB_STAT_EIG_NOCONV);
}

// The rows above a panel (kb_eig.hpp, hess_ytop_block): Y(r, :) = A0(r, p0+1:) VT(p0+1:, :) for r <= p0 as FP64-MFMA tiles
// (one workgroup = 64 rows x NB outputs, k = the columns right of p0; half of the tile's columns multiply zeros), then the
// same rows of the panel's own columns 1 .. NB-1.  A workgroup reads and writes its own rows only.  (Tiles of k_hess_z's launch.)
__device__ __forceinline__ void hess_ytop_tile(const KbItem& it, cd* arena, int panel, int tile, kb_tu_stage* s_op) {
    const int n = it.l;
    const int p0 = panel * KB_NB;
    const int r0 = tile * 64;
    if (r0 > p0) return;
    cd* W = arena + it.off[KB_BUF_P];
    cd* Y = arena + it.off[KB_BUF_Q];
    const cd* VT = arena + it.off[KB_BUF_H] + (size_t)n * KB_NB + KB_NB * KB_NB;
    const int K1 = n - p0 - 1, K1p = (K1 + KB_TU_KC - 1) / KB_TU_KC * KB_TU_KC;
    mfma_tile_ks2<true, false, true>(
        s_op,
        [&](int i, int k) -> cd {
            const int r = r0 + i;
            return (r <= p0 && k < K1) ? W[r + (size_t)(p0 + 1 + k) * n] : czero();
        },
        [&](int i, int k) -> cd { return (i < KB_NB && k < K1) ? conj(VT[(p0 + 1 + k) + (size_t)i * n]) : czero(); },
        [&](int i, int jx) -> cd* {
            const int r = r0 + i;
            return (r <= p0 && jx < KB_NB) ? &Y[r + (size_t)jx * n] : nullptr;
        },
        K1p / KB_TU_KC);
    __threadfence_block();
    __syncthreads();                                          // (the workgroup reads back its own rows of Y)
    for (int e = threadIdx.x; e < 64 * (KB_NB - 1); e += 256) {
        const int r = r0 + (e & 63), j = 1 + (e >> 6);
        if (r > p0) continue;
        cd acc = czero();
        for (int t = 0; t < j; ++t) acc = acc + Y[r + (size_t)t * n] * conj(hess_vt(W, n, p0, p0 + j, t));
        W[r + (size_t)(p0 + j) * n] = W[r + (size_t)(p0 + j) * n] - acc;
    }
}

// Deferred left factor of a panel, all CUs:  Z(c, :) = A0(:, c)^H VT - V(c, :) MT  for the columns right of the
// panel (kb_eig.hpp, hess_z_block).  One workgroup = 64 columns x NB outputs; A0 and VT go through LDS in chunks
// of 16 rows (A0 is read once per panel here instead of once per column inside the panel).
// Both products behind a panel in ONE launch: tiles [0, nzt) are Z's (64 columns each), the rest the rows above the panel.
__global__ void __launch_bounds__(256) k_hess_z(const KbItem* __restrict__ items, const int* __restrict__ perm,
                                                 cd* arena, int panel, int nzt) {
    const KbItem it = items[perm[blockIdx.y]];
    const int n = it.l;
    if (panel >= bidiag_num_panels(n)) return;
    __shared__ kb_tu_stage s_op[2];
    if ((int)blockIdx.x >= nzt) {
        hess_ytop_tile(it, arena, panel, blockIdx.x - nzt, s_op);
        return;
    }
    const int p0 = panel * KB_NB;
    const int cbase = p0 + KB_NB;
    const int c0 = cbase + blockIdx.x * 64;
    if (c0 >= n) return;
    const cd* W = arena + it.off[KB_BUF_P];
    cd* Z = arena + it.off[KB_BUF_H];
    const cd* MT = Z + (size_t)n * KB_NB;
    const cd* VT = MT + KB_NB * KB_NB;
    // Z[c, t] = sum_r conj(A0[r, c]) VT[r, t] - sum_u V(c, u) MT[u, t]  as ONE FP64-MFMA tile product over k = (the rows
    // r = p0 + 1 .. n - 1, padded to a multiple of 8) ++ (u = 0 .. NB - 1):  Aop(c, k) = conj(A0[r, c]) | -V(c, u),
    // Bop(t, k) = conj(VT[r, t]) | conj(MT[u, t])  (the tile computes sum_k Aop conj(Bop)); both operands are contiguous
    // along r: k-fastest staging.  Only 32 of the tile's 64 columns exist (t < NB): half the MFMAs multiply zeros, still
    // several times the rate of the vector-FMA tiles this replaces (3.3 TFLOP/s on the C4 batch).
    const int K1 = n - p0 - 1, K1p = (K1 + KB_TU_KC - 1) / KB_TU_KC * KB_TU_KC;
    mfma_tile_ks2<true, true, true>(
        s_op,
        [&](int i, int k) -> cd {
            const int c = c0 + i;
            if (c >= n) return czero();
            if (k < K1p) { const int r = p0 + 1 + k; return (r < n) ? conj(W[r + (size_t)c * n]) : czero(); }
            return -hess_vt(W, n, p0, c, k - K1p);           // V(c, u) (the head of the last reflector sits at c = cbase)
        },
        [&](int i, int k) -> cd {
            if (i >= KB_NB) return czero();
            if (k < K1p) { const int r = p0 + 1 + k; return (r < n) ? conj(VT[r + (size_t)i * n]) : czero(); }
            return conj(MT[(k - K1p) + i * KB_NB]);
        },
        [&](int i, int jx) -> cd* {
            const int c = c0 + i;
            return (c < n && jx < KB_NB) ? &Z[c + (size_t)jx * n] : nullptr;
        },
        (K1p + KB_NB) / KB_TU_KC);
}

__global__ void __launch_bounds__(256) k_hess_update(const KbItem* __restrict__ items, const int* __restrict__ perm,
                                                      cd* arena, int panel) {
    const KbItem it = items[perm[blockIdx.z]];
    const int n = it.l;
    if (panel >= bidiag_num_panels(n)) return;
    const int p0 = panel * KB_NB;
    const int cbase = p0 + KB_NB;                 // first updated column
    const int ncol = n - cbase;
    if (blockIdx.x * 64 >= n || blockIdx.y * 64 >= ncol) return;
    const int r0 = blockIdx.x * 64;                       // global row of the tile origin
    const int c0 = cbase + blockIdx.y * 64;               // global column
    cd* W = arena + it.off[KB_BUF_P];
    const cd* Y = arena + it.off[KB_BUF_Q];
    const cd* Z = arena + it.off[KB_BUF_H];
    // Aop = [Y | V],  Bop = [V | Z]
    mfma_rank2nb_tile(
        [&](int i, int k) -> cd {
            const int r = r0 + i;
            if (r >= n) return czero();
            return (k >= KB_NB) ? hess_vt(W, n, p0, r, k - KB_NB) : Y[r + (size_t)k * n];
        },
        [&](int i, int k) -> cd {
            const int c = c0 + i;
            if (c >= n) return czero();
            return (k >= KB_NB) ? Z[c + (size_t)(k - KB_NB) * n] : hess_vt(W, n, p0, c, k);
        },
        [&](int i, int jx) -> cd* {
            const int r = r0 + i, c = c0 + jx;
            return (r < n && c < n) ? &W[r + (size_t)c * n] : nullptr;
        });
}

__global__ void __launch_bounds__(1024) k_hess(const KbItem* __restrict__ items, const int* __restrict__ perm,
                                                cd* arena, double* varena, int smem_bytes, int blocked) {
    const KbItem it = items[perm[blockIdx.x]];
    const DevCtx ctx = make_ctx(smem_bytes);
    const int n = it.l;
    cd* W = arena + it.off[KB_BUF_P];
    cd* Hc = arena + it.off[KB_BUF_H];
    double* dv = varena + it.voff;
    cd* tauh = reinterpret_cast<cd*>(dv + KB_V_TAUQ * it.vstride);   // tauq is dead by now
    gehd2(ctx, n, W, n, tauh, blocked ? bidiag_num_panels(n) * KB_NB : 0);
    hess_copy(ctx, n, W, n, Hc, n);
    double rmax = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double r = 0.0;
        const int jlo = i > 0 ? i - 1 : 0;
        for (int j = jlo; j < n; ++j) r += cabs(Hc[i + (size_t)j * n]);
        rmax = fmax(rmax, r);
    }
    rmax = ctx.block_max(rmax);
    if (threadIdx.x == 0) dv[KB_V_MISC * it.vstride] = rmax;
}

// QR iteration (kb_hqr2.hpp): double-shift bulges, register-systolic strip replay, window + log in LDS.
// k_hqr2: queue == nullptr: workgroup b solves member perm[b]; otherwise the workgroups of the launch take members
// perm[0], perm[1], ... (largest first) from the queue until it is empty.
// k_hqr2_team (large members): workgroup 2t is the chase workgroup of member t, workgroup 2t + 1 its helper on
// another CU (far strip units).  Team-major numbering: a team's two workgroups are dispatched together.  The host
// launches at most as many teams as fit the chip at one workgroup per CU, so every workgroup of the launch is
// resident; all waits are bounded (abort flag + status bit).
#include "kbdm_ab_kernels.hpp"

// needqr != nullptr: only the members flagged there are solved (the fallback behind the Ehrlich-Aberth path).
__global__ void __launch_bounds__(512) k_hqr2(const KbItem* __restrict__ items, const int* __restrict__ perm,
                                                   cd* arena, cd* mu_out, int* status, int smem_bytes, int nbmax,
                                                   int win_w, MsStats* prof, int count, int* queue, const int* needqr) {
    const DevCtx ctx = make_ctx(smem_bytes);
    __shared__ int info;
    __shared__ int next;
    for (;;) {
        int idx = blockIdx.x;
        if (queue) {
            if (threadIdx.x == 0) next = atomicAdd(queue, 1);
            __syncthreads();
            idx = next;
            __syncthreads();
        }
        if ((unsigned)idx >= (unsigned)count) break;
        const int item = perm[idx];
        if (needqr && !needqr[item]) {
            if (!queue) break;
            __syncthreads();
            continue;
        }
        const KbItem it = items[item];
        cd* Hc = arena + it.off[KB_BUF_H];
        cd* mu = mu_out + it.line_off;
        hqr2_eigvals(ctx, it.l, Hc, it.l, mu, &info, nbmax, win_w, prof ? prof + item : nullptr);
        if (threadIdx.x == 0 && info != 0) atomicOr(&status[item], KB_STAT_EIG_NOCONV);
        if (!queue) break;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(512) k_hqr2_team(const KbItem* __restrict__ items, const int* __restrict__ perm,
                                                        cd* arena, cd* mu_out, int* status, int smem_bytes, int nbmax,
                                                        int win_w, TeamCtl* ctl, char* rings, MsStats* prof) {
    const int team = blockIdx.x >> 1, role = blockIdx.x & 1;
    const int item = perm[team];
    const KbItem it = items[item];
    const DevCtx ctx = make_ctx(smem_bytes);
    cd* Hc = arena + it.off[KB_BUF_H];
    Team2<DevCtx> tm;
    tm.ctl = ctl + item;
    tm.rec_bytes = team2_rec_bytes(win_w);
    tm.ring = rings + (size_t)item * KB_TEAM_SLOTS * tm.rec_bytes;
    tm.g = 0; tm.g_batch = 0; tm.failed = 0;
    tm.A = HSc1::make(Hc, it.l, it.l);
    tm.W = win_w;
    if (role == 0) {
        cd* mu = mu_out + it.line_off;
        __shared__ int info;
        hqr2_eigvals(ctx, it.l, Hc, it.l, mu, &info, nbmax, win_w, prof ? prof + item : nullptr, &tm);
        if (threadIdx.x == 0 && info != 0) atomicOr(&status[item], KB_STAT_EIG_NOCONV);
    } else {
        team2_helper_main(ctx, tm);
    }
}

__global__ void __launch_bounds__(1024) k_invit(const KbItem* __restrict__ items, const int* __restrict__ perm,
                                                 cd* arena, double* varena, const cd* mu_out, int* status,
                                                 int smem_bytes, int nmin) {
    const int item = perm[blockIdx.x];
    const KbItem it = items[item];
    if (it.l <= nmin) return;
    const DevCtx ctx = make_ctx(smem_bytes);
    const int n = it.l;
    const cd* Hw = arena + it.off[KB_BUF_P];     // H above the Householder vectors of k_hess
    cd* X = arena + it.off[KB_BUF_H];
    const double hnorm = varena[it.voff + KB_V_MISC * it.vstride];
    int nw = ctx.scratch_bytes() / invit_scratch_bytes_per_wave(n);
    if (nw > ctx.nwaves()) nw = ctx.nwaves();
    __shared__ int weak;
    if (threadIdx.x == 0) weak = 0;
    __syncthreads();
    invit<DevCtx, 8>(ctx, n, Hw, n, mu_out + it.line_off, hnorm, X, n, nw, &weak, blockIdx.y, gridDim.y);
    __syncthreads();
    if (threadIdx.x == 0 && weak) atomicOr(&status[item], KB_STAT_INVIT_WEAK);
}

// Register-resident form (members with l <= 64 MAXC): no LDS, 512-thread workgroups, every wavefront solves
// the eigenvalues kk = (global wavefront index), + (wavefronts per member), ...
template <int MAXC>
__global__ void __launch_bounds__(512) k_invit_reg(const KbItem* __restrict__ items, const int* __restrict__ perm,
                                                    cd* arena, double* varena, const cd* mu_out, int* status, int nmin) {
    const int item = perm[blockIdx.x];
    const KbItem it = items[item];
    const int n = it.l;
    // a member is always solved by the SAME instantiation (the smallest MAXC that holds it), whatever else is in the
    // batch: the host launches one kernel per size class, each takes the members with nmin < l <= 64 MAXC
    if (n > MAXC * 64 || n <= nmin) return;
    const DevCtx ctx = make_ctx(0);
    const cd* Hw = arena + it.off[KB_BUF_P];
    cd* X = arena + it.off[KB_BUF_H];
    const double hnorm = varena[it.voff + KB_V_MISC * it.vstride];
    __shared__ int weak;
    if (threadIdx.x == 0) weak = 0;
    __syncthreads();
    const int wpb = blockDim.x >> 6;
    // dynamic LDS (kb_smem): wpb x MAXC x 64 multipliers
    invit_reg<MAXC>(ctx, n, Hw, n, mu_out + it.line_off, hnorm, X, n, (int)blockIdx.y * wpb + ctx.wave(),
                    (int)gridDim.y * wpb, &weak, reinterpret_cast<cd*>(kb_smem) + (size_t)ctx.wave() * MAXC * 64);
    __syncthreads();
    if (threadIdx.x == 0 && weak) atomicOr(&status[item], KB_STAT_INVIT_WEAK);
}

// The same for members of up to MAXC * 64 = 1280 rows (STREAM form: see InvitRegState); no LDS.
template <int MAXC>
__global__ void __launch_bounds__(256) k_invit_big(const KbItem* __restrict__ items, const int* __restrict__ perm,
                                                    cd* arena, double* varena, const cd* mu_out, int* status, int nmin) {
    const int item = perm[blockIdx.x];
    const KbItem it = items[item];
    const int n = it.l;
    if (n > MAXC * 64 || n <= nmin) return;          // (size classes: see k_invit_reg)
    const DevCtx ctx = make_ctx(0);
    const cd* Hw = arena + it.off[KB_BUF_P];
    cd* X = arena + it.off[KB_BUF_H];
    const double hnorm = varena[it.voff + KB_V_MISC * it.vstride];
    __shared__ int weak;
    if (threadIdx.x == 0) weak = 0;
    __syncthreads();
    const int wpb = blockDim.x >> 6;
    // dynamic LDS (kb_smem): wpb x MAXC x 64 multipliers
    invit_reg<MAXC, true>(ctx, n, Hw, n, mu_out + it.line_off, hnorm, X, n, (int)blockIdx.y * wpb + ctx.wave(),
                    (int)gridDim.y * wpb, &weak, nullptr);
    __syncthreads();
    if (threadIdx.x == 0 && weak) atomicOr(&status[item], KB_STAT_INVIT_WEAK);
}

// ------------------------------------------------------------------------------------
// Epilogue (kbdm.py:71-90 + sampling.py:75-97).  One wavefront per spectral line k:
//   N_k = sum_i B[i,k] T[i,k]   (bilinear, unconjugated)      D_sqrt = c[:m] . b_k / sqrt(N_k)
//   D = D_sqrt^2 = (c.b)^2 / N,  A = |D|, PH = arg D,  F = arg(mu)/(2 pi dwell), T2 = -dwell/ln|mu|
__global__ void __launch_bounds__(256) k_epilogue(const KbItem* __restrict__ items, const int* __restrict__ perm,
                                                   const cd* __restrict__ signals, int N, const cd* arena,
                                                   const cd* mu_all, double dwell, double* lines,
                                                   unsigned char* keep) {
    const KbItem it = items[perm[blockIdx.y]];
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (k >= it.l) return;
    const int lane = threadIdx.x & 63;
    const int m = it.m;
    const cd* Bm = arena + it.off[KB_BUF_A] + (size_t)k * m;
    const cd* T = arena + it.off[KB_BUF_Q] + (size_t)k * m;
    const cd* sig = signals + (size_t)it.sig * N;
    cd nk = czero(), ds = czero();
    for (int i = lane; i < m; i += 64) {
        const cd b = Bm[i];
        cfma(nk, b, T[i]);
        cfma(ds, sig[i], b);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        nk.x += __shfl_xor(nk.x, o, 64); nk.y += __shfl_xor(nk.y, o, 64);
        ds.x += __shfl_xor(ds.x, o, 64); ds.y += __shfl_xor(ds.y, o, 64);
    }
    if (lane == 0) {
        const cd mu = mu_all[it.line_off + k];
        const cd D = cdiv(ds * ds, nk);
        const double A = hypot(D.x, D.y);
        const double PH = atan2(D.y, D.x);
        const double lnabs = log(hypot(mu.x, mu.y));
        const double T2 = -dwell / lnabs;
        const double F = atan2(mu.y, mu.x) / (6.283185307179586476925286766559 * dwell);
        double* out = lines + 4 * (it.line_off + k);
        out[0] = A; out[1] = T2; out[2] = F; out[3] = PH;
        keep[it.line_off + k] = (A > 1e-6 && T2 > 0.0) ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------
// Layout helpers for the stage entry points (row-major host matrices <-> column-major).
__global__ void k_transpose_in(const KbItem* __restrict__ items, const cd* __restrict__ src, cd* arena, int buf,
                               int use_l) {
    const KbItem it = items[blockIdx.y];
    const int n = use_l ? it.l : it.m;
    const cd* s = src + it.hk_off;
    cd* d = arena + it.off[buf];
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n * n; e += gridDim.x * blockDim.x) {
        const int i = e % n, j = e / n;
        d[i + (size_t)j * n] = s[(size_t)i * n + j];
    }
}
__global__ void k_transpose_out(const KbItem* __restrict__ items, const cd* arena, int buf, cd* dst, int use_l) {
    const KbItem it = items[blockIdx.y];
    const int n = use_l ? it.l : it.m;
    const cd* s = arena + it.off[buf];
    cd* d = dst + it.hk_off;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n * n; e += gridDim.x * blockDim.x) {
        const int j = e % n, i = e / n;
        d[(size_t)i * n + j] = s[i + (size_t)j * n];
    }
}
