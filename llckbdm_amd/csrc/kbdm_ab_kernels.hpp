// gfx950 kernels of the divide-and-conquer Ehrlich-Aberth eigenvalue solver (algorithm: kb_aberth.hpp; the fast path
// of the reference's scipy.linalg.eig at kbdm.py:192, eigenvalues only; eigenvectors follow by inverse iteration).
//
//   k_ab_leaf    one wavefront per (member, leaf <= 32 rows): the small solver of kb_hqr2.hpp (hqr2_shifts); workgroup 0
//                of a member also scans the subdiagonal (a negligible entry sends the member to the QR iteration)
//   k_ab_iter    one workgroup per (member, node of the step's level, tile of 64 roots): ONE Aberth iteration of those
//                roots: the repulsion sums, Hyman's recurrence for the 64 roots and their derivatives as a blocked
//                product H X (FP64 MFMA for the part of a 32-row block that multiplies finished rows, a two-wavefront
//                recurrence for the 32 x 32 triangle), the update.  Launched KB_AB_BUDGET times per level; a tile
//                whose roots have all settled only copies them through.
//   k_ab_finish  per member: every root settled? power sums against trace(H), trace(H^2); writes mu or flags the
//                member for the QR iteration (k_hqr2 then runs for flagged members only)
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "kb_aberth.hpp"
#include "kbdm_device.h"

// (kb_smem, make_ctx, kb_d4 come from kbdm_kernels.hpp, which includes this file after kb_hqr2.hpp)

constexpr int KB_AB_KC = 8;                     // k per staged chunk of the block product
constexpr int KB_AB_TP = 33;                    // pitch (cd) of a wavefront's transposition area: 16 rows x 32 columns
constexpr int KB_AB_PA = 48;                    // LDS pitch (doubles) of the H chunk rows: 2 * pitch = 32 mod 64 banks, so the two k rows
                                                // of a half-wavefront's ds_read_b64 fall on disjoint banks

struct AbLds {                                  // dynamic LDS of k_ab_iter
    double stage[4][2][2][KB_AB_KC][KB_AB_PA];  // [wavefront][H chunk | panel chunk][re|im][k][row / column]: private staging areas
    kb::cd ht[KB_AB_BLK][KB_AB_BLK + 1];        // the block's triangle of H (row k, column j), padded
    kb::cd inv[KB_AB_BLK];                      // 1 / H[k, k-1]
    kb::cd prow[2 * KB_AB_TILE];                // the finished row above the block (x_{k_hi}, y_{k_hi}); at the end rho, rho'
    double fac[2 * KB_AB_TILE];                 // power-of-two rescaling of a column at the end of a block
    kb::cd z[KB_AB_TILE];
    kb::cd S[KB_AB_TILE];
    double part[4][KB_AB_TILE][2];
    int cnt[256];                               // unsettled roots per thread of the flag scan
    int ridx[KB_AB_TILE];                       // node-local index of this tile's roots
    int flags[8];
};

// every lane gets the value of lane (lane & 31)
__device__ __forceinline__ double ab_lower_half(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]);
}
// every lane gets the value of lane (lane | 32)
__device__ __forceinline__ double ab_upper_half(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[1], (int)a[1]);
}
// lanes 16-31 / 48-63 get the value of the lane 16 below, the others keep theirs (v_permlane16_swap)
__device__ __forceinline__ double ab_row_below(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]);
}
// LDS access through a 32-bit LDS address + byte offset (see the triangle of k_ab_iter)
typedef double ab_d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) ab_d2 ab_lds_d2;
__device__ __forceinline__ unsigned ab_lds_addr(const void* p) { return (unsigned)(size_t)p; }     // flat -> LDS: the low 32 bits
__device__ __forceinline__ kb::cd ab_lds_ld(unsigned base, int off) {
    const ab_d2 v = *reinterpret_cast<const ab_lds_d2*>((size_t)(base + (unsigned)off));
    return kb::mk(v.x, v.y);
}
__device__ __forceinline__ void ab_lds_st(unsigned base, int off, kb::cd v) {
    *reinterpret_cast<ab_lds_d2*>((size_t)(base + (unsigned)off)) = (ab_d2){v.x, v.y};
}

__device__ __forceinline__ kb::AbWs ab_item_ws(const kb::KbItem& it, double* dcarena) { return kb::ab_ws(dcarena + it.dc_off, it.l); }

// column of the panel that holds root r's x (which = 0) / derivative (which = 1): a wavefront owns 16 roots with both
// (its 32 columns through the whole recurrence: block product, triangle, panel rows)
__device__ __forceinline__ int ab_col(int r, int which) { return (r >> 4) * 32 + which * 16 + (r & 15); }

// Sub-node (start, size) at depth d of a leaf of n rows that holds position j (the same halving as ab_node)
__device__ __forceinline__ void ab_subnode(int n, int d, int j, int& sa, int& sn) {
    sa = 0; sn = n;
    for (int b = 0; b < d; ++b) {
        const int h = sn / 2;
        if (j < sa + h) sn = h; else { sa += h; sn -= h; }
    }
}

// One wavefront per (member, leaf of <= 32 rows): the same divide and conquer INSIDE the leaf, down to 1 x 1 / 2 x 2
// blocks (closed form), with the whole recurrence in LDS: lane r < 32 carries x of root r, lane r + 32 its derivative.
__global__ void __launch_bounds__(64) k_ab_leaf(const kb::KbItem* __restrict__ items, const int* __restrict__ perm, kb::cd* arena,
                                                 const double* __restrict__ varena, double* dcarena, int* needqr, int smem_bytes) {
    using namespace kb;
    const int item = perm[blockIdx.y];
    const KbItem it = items[item];
    const int l = it.l, D = ab_depth(l);
    if ((int)blockIdx.x >= (1 << D)) return;
    const DevCtx ctx = make_ctx(smem_bytes);
    const cd* H = arena + it.off[KB_BUF_H];
    const AbWs ws = ab_item_ws(it, dcarena);
    const double hnorm = varena[it.voff + KB_V_MISC * it.vstride];
    const int t = threadIdx.x;
    if (blockIdx.x == 0) {                      // the subdiagonal scan
        int bad = 0;
        for (int k = 1 + t; k < l; k += 64)
            if (ab_negligible_sub(H[k + (size_t)(k - 1) * l], H[k + (size_t)k * l], H[(k - 1) + (size_t)(k - 1) * l])) bad = 1;
        bad = ctx.block_max(bad);
        if (t == 0 && bad) atomicOr(&needqr[item], 1);
    }
    const AbNode nd = ab_node(l, D, blockIdx.x);
    const int n = nd.n, a = nd.a;
    __shared__ cd Hl[KB_AB_LEAF][KB_AB_LEAF + 1];
    __shared__ cd XY[2][KB_AB_LEAF][KB_AB_LEAF + 1];       // [x | y][row k][root]
    __shared__ cd inv[KB_AB_LEAF], zl[KB_AB_LEAF], rr[2][KB_AB_LEAF];
    for (int idx = t; idx < KB_AB_LEAF * KB_AB_LEAF; idx += 64) {
        const int r = idx & 31, c = idx >> 5;
        Hl[r][c] = (r < n && c < n && r <= c + 1) ? H[(a + r) + (size_t)(a + c) * l] : czero();
    }
    __syncthreads();
    if (t < KB_AB_LEAF) inv[t] = (t >= 1 && t < n) ? ab_recip(Hl[t][t - 1]) : czero();
    const int root = t & 31, which = t >> 5;
    // depth of the inner tree: halve until every block has at most two rows
    int dl = 0;
    for (int s2 = n; s2 > 2; s2 = s2 - s2 / 2) ++dl;
    int sa, sn;
    ab_subnode(n, dl, root < n ? root : 0, sa, sn);
    if (which == 0 && root < n) {               // closed forms
        cd z1 = Hl[sa][sa], z2 = z1;
        if (sn == 2) eig2x2(Hl[sa][sa], Hl[sa][sa + 1], Hl[sa + 1][sa], Hl[sa + 1][sa + 1], z1, z2);
        zl[root] = (root == sa) ? z1 : z2;
    }
    __syncthreads();
    int failed = 0;
    for (int d = dl - 1; d >= 0; --d) {
        ab_subnode(n, d, root < n ? root : 0, sa, sn);
        const bool live = root < n;
        const bool strict = D == 0 && d == 0;            // only the root of the whole tree is a result; everything else starts its parent
        cd z = live ? ab_perturb(zl[root], a + root, hnorm) : czero();
        __syncthreads();
        if (which == 0 && live) zl[root] = z;
        __syncthreads();
        bool settled = !live;
        double lastdz = 0.0;
        const int snmax = (n >> d) + 1;                       // (an upper bound of the node sizes of this depth)
        for (int iter = 0; iter < (strict ? 40 : KB_AB_INNER_BUDGET); ++iter) {
            if (__ballot(!settled) == 0ull) break;
            // Hyman's recurrence of this lane's root in its node: rows sn-1 .. 0
            if (live) XY[which][sn - 1][root] = which ? czero() : mk(1.0, 0.0);
            cd rho = czero();
            for (int k = snmax - 1; k >= 0; --k) {
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (live && k < sn) {
                    cd s = czero(), s1 = czero();
                    int j = k;
                    for (; j + 1 < sn; j += 2) {
                        cfma(s, Hl[sa + k][sa + j], XY[which][j][root]);
                        cfma(s1, Hl[sa + k][sa + j + 1], XY[which][j + 1][root]);
                    }
                    if (j < sn) cfma(s, Hl[sa + k][sa + j], XY[which][j][root]);
                    s = s + s1;
                    s = s - z * XY[which][k][root];
                    if (which) s = s - XY[0][k][root];
                    if (k >= 1) XY[which][k - 1][root] = -(s * inv[sa + k]);
                    else rho = s;
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (live) rr[which][root] = rho;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            cd zn = z;
            if (which == 0 && live && !settled) {
                cd S = czero();
                for (int j = sa; j < sa + sn; ++j)
                    if (j != root) S = S + ab_recip(z - zl[j]);
                double dz;
                zn = ab_update(z, rr[0][root], rr[1][root], S, &dz);
                lastdz = dz;
                if (strict ? ab_converged(dz, zn, hnorm) : ab_converged_inner(dz, zn, hnorm)) settled = true;
            }
            __builtin_amdgcn_wave_barrier();
            if (which == 0 && live) zl[root] = zn;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            z = live ? zl[root] : czero();
            // the derivative lane follows its root's state
            const int st_x = __shfl(settled ? 1 : 0, root, 64);
            if (which) settled = st_x != 0;
        }
        // (below the root of the whole tree an unsettled root is only a worse starting value; a value that is not finite is not)
        if (which == 0 && live && !(strict ? ab_acceptable(lastdz, z, hnorm) : ab_finite(z))) failed = 1;
    }
    failed = ctx.block_max(failed);
    if (t < n) {
        ws.z[0][a + t] = zl[t];
        ws.conv[0][a + t] = 1;
        ws.lastc[a + t] = 0.0;
    }
    if (t == 0 && failed) atomicOr(&needqr[item], 1);
}

// grid (tiles * nodes of the deepest level of this step, members); block 256; dynamic LDS sizeof(AbLds)
__global__ void __launch_bounds__(256) k_ab_iter(const kb::KbItem* __restrict__ items, const int* __restrict__ perm,
                                                  const kb::cd* __restrict__ arena, const double* __restrict__ varena, double* dcarena,
                                                  const int* __restrict__ needqr, int step, int iter, int* abstat, int dbg) {
    using namespace kb;
    const int item = perm[blockIdx.y];
    if (needqr[item]) return;
    const KbItem it = items[item];
    const int l = it.l, depth = ab_depth(l) - 1 - step;
    if (depth < 0) return;
    const bool tail_on = (dbg & 16) == 0;                   // (the host sets bit 16 when k_ab_tail is not launched)
    const int Tl = (ab_level_nmax(l, depth) + KB_AB_TILE - 1) / KB_AB_TILE;
    const int idx = blockIdx.x / Tl, tile = blockIdx.x % Tl;
    if (idx >= (1 << depth)) return;
    const AbNode nd = ab_node(l, depth, idx);
    const int n = nd.n, a0 = nd.a;
    if (tile * KB_AB_TILE >= n) return;
    AbLds& L = *reinterpret_cast<AbLds*>(kb_smem);
    const AbWs ws = ab_item_ws(it, dcarena);
    const int bin = (step * KB_AB_BUDGET + iter) & 1;
    const cd* zin = ws.z[bin] + a0;
    cd* zout = ws.z[bin ^ 1] + a0;
    const int* cin = ws.conv[bin] + a0;
    int* cout = ws.conv[bin ^ 1] + a0;
    const cd* H = arena + it.off[KB_BUF_H];
    const double hnorm = varena[it.voff + KB_V_MISC * it.vstride];
    const int t = threadIdx.x;
    // ---- the unsettled roots of the node, compacted: tile k takes the 64 k-th .. of them.  Every workgroup of the node
    // scans the node's flags (of the previous iteration: double buffered, so all of them see the same set) and carries
    // the settled roots through - identical values from every workgroup.  The first iteration of a level re-opens every
    // root (the children's eigenvalues are only starting values) and separates coincident ones.
    const int per = (n + 255) / 256;
    const int j0 = t * per, j1 = (j0 + per < n) ? j0 + per : n;
    {
        int c = 0;
        // below the root of the tree a level stops after KB_AB_INNER_BUDGET iterations whatever has settled: its eigenvalues
        // only start the parent (measured on C2: the inner levels' work falls to less than half, the root level's grows by 8 %)
        const bool capped = depth > 0 && iter >= KB_AB_INNER_BUDGET;
        for (int j = j0; j < j1; ++j) {
            if ((iter != 0 && cin[j]) || capped) { zout[j] = zin[j]; cout[j] = 1; }
            else ++c;
        }
        L.cnt[t] = c;
        if (t == 0) L.flags[1] = 0;
    }
    __syncthreads();
    int rank = 0;
    for (int u = 0; u < t; ++u) rank += L.cnt[u];
    if (t == 255) L.flags[0] = rank + L.cnt[255];
    for (int j = j0; j < j1; ++j)
        if ((iter == 0 || !cin[j]) && !(depth > 0 && iter >= KB_AB_INNER_BUDGET)) {
            const int q = rank - tile * KB_AB_TILE;
            if (q >= 0 && q < KB_AB_TILE) L.ridx[q] = j;
            ++rank;
        }
    __syncthreads();
    const int nactive = L.flags[0];
    if (depth == 0 && tile == 0 && t == 0) *ws.nact = nactive;         // (k_ab_tail's workgroups decide on it without a scan)
    if (tail_on && ab_tail_takes(l, depth, iter, nactive)) return;     // the few roots left get a wavefront each (k_ab_tail)
    const int nr = (nactive - tile * KB_AB_TILE < KB_AB_TILE) ? nactive - tile * KB_AB_TILE : KB_AB_TILE;
    if (nr <= 0) return;                                    // this tile has nothing left to iterate
    if (abstat && t == 0) atomicAdd(&abstat[step * KB_AB_BUDGET + iter], 1);
    // (diagnostic, dbg & 8: 10 ns ticks per phase of the full tiles of nodes with more than 256 rows, rows 8.. of abstat)
    const bool prof = (dbg & 8) && abstat && t == 0 && nr == KB_AB_TILE && n > 256;
    long long tp = prof ? wall_clock64() : 0;
    auto lap = [&](int ph) {
        if (!prof) return;
        const long long now = wall_clock64();
        atomicAdd(&abstat[8 * KB_AB_BUDGET + ph], (int)(now - tp));
        tp = now;
    };
    // Panel P[row][128] of this tile in global memory.  In the first iteration of a level its first rows hold the separated
    // starting values of the whole node (computed once per tile instead of once per pair of roots: ab_perturb costs a sincos)
    cd* P = ws.panel + ((size_t)a0 * Tl + (size_t)tile * n) * (2 * KB_AB_TILE);
    const cd* zsrc = zin;
    if (iter == 0) {
        for (int j = t; j < n; j += 256) P[j] = ab_perturb(zin[j], a0 + j, hnorm);
        zsrc = P;
        __syncthreads();
    }
    if (t < KB_AB_TILE) L.z[t] = (t < nr) ? zsrc[L.ridx[t]] : mk(0.0, 0.0);
    __syncthreads();
    // ---- repulsion sums S_i = sum_{j != i} 1 / (z_i - z_j) over the node (four threads per root).  S only steers the
    // iteration (its fixed points are the roots whatever S is): hardware reciprocal + one Newton step instead of divisions
    {
        const int i = t & 63, q = t >> 6;
        double sx = 0.0, sy = 0.0;
        if (i < nr) {
            const cd zi = L.z[i];
            const int me = L.ridx[i];
            for (int j = q; j < n; j += 4) {
                if (j == me) continue;
                const cd d = zi - zsrc[j];
                const double den = d.x * d.x + d.y * d.y;
                double inv = __builtin_amdgcn_rcp(den);
                inv = inv * fma(-den, inv, 2.0);
                sx = fma(d.x, inv, sx); sy = fma(-d.y, inv, sy);
            }
        }
        L.part[q][i][0] = sx; L.part[q][i][1] = sy;
    }
    __syncthreads();
    if (t < KB_AB_TILE)
        L.S[t] = mk((L.part[0][t][0] + L.part[1][t][0]) + (L.part[2][t][0] + L.part[3][t][0]),
                    (L.part[0][t][1] + L.part[1][t][1]) + (L.part[2][t][1] + L.part[3][t][1]));
    // ---- Hyman's recurrence, rows n-1 .. 0, blocks of 32 rows
    const int wave = t >> 6, lane = t & 63;
    const int li = lane & 15, lk = lane >> 4;
    const bool wact = wave * 16 < nr;                       // this wavefront's 16 roots: any of them live?
    if (t < 2 * KB_AB_TILE) L.fac[t] = 1.0;
    if (t < 2 * KB_AB_TILE) {                               // row n-1: x = 1, y = 0
        const int c = t, which = (c >> 4) & 1;
        const cd v = which ? czero() : mk(1.0, 0.0);
        P[(size_t)(n - 1) * (2 * KB_AB_TILE) + c] = v;
        L.prow[c] = v;
    }
    __syncthreads();
    int k_hi = n - 1;
    lap(0);
    while (k_hi >= 0) {
        // slot r of the block <-> row k = kb0s + r of the node, r = 0 .. 31; the topmost block has kb0s <= 0 and its
        // slots below rmin are empty (k < 0)
        const int kb0s = k_hi - (KB_AB_BLK - 1);
        const int rmin = kb0s < 0 ? -kb0s : 0;
        // -- the block's triangle of H and the subdiagonals: loaded now, written to LDS after the block product (the loads
        // travel under the product instead of in front of it)
        cd htr[KB_AB_BLK * KB_AB_BLK / 256], hsub = mk(1.0, 0.0);
#pragma unroll
        for (int i = 0; i < KB_AB_BLK * KB_AB_BLK / 256; ++i) {
            const int e = t + 256 * i, r = e & 31, j = e >> 5;
            const int rc = r >= rmin ? r : rmin, jc = j >= rmin ? j : rmin;
            htr[i] = H[(a0 + kb0s + rc) + (size_t)(a0 + kb0s + jc) * l];
        }
        if (t < KB_AB_BLK) {
            const int k = kb0s + t;
            hsub = H[(a0 + (k >= 1 ? k : 1)) + (size_t)(a0 + (k >= 1 ? k : 1) - 1) * l];
        }
        // -- G[k] = sum_{j = k_hi}^{n-1} H[k, j] P[j, :]   (rows k of the block x 128 columns, MFMA)
        kb_d4 acc_re[2][2], acc_im[2][2];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) { acc_re[x][y] = (kb_d4){0, 0, 0, 0}; acc_im[x][y] = (kb_d4){0, 0, 0, 0}; }
        const int K = n - k_hi;                              // stored rows j = k_hi .. n-1
        const int nch = (K + KB_AB_KC - 1) / KB_AB_KC;
        // Chunks of 8 rows j from the OLDEST (j = n-1 ..) to the newest (the rows the previous block has just stored come
        // last).  Every wavefront runs its OWN pipeline - no workgroup barrier inside the product: it stages the H chunk
        // (32 rows x 8 k; four copies per workgroup, out of the L1) and the 32 panel columns it multiplies into a private LDS
        // area (one buffer: the DS operations of a wavefront execute in order), with the global loads two chunks ahead in two
        // register sets.  Loads are unconditional (chunk index and rows clamped into the stored rows, the value masked when it
        // is written to LDS): no branch between a load and its use, so the compiler counts the outstanding loads.
        // lane -> (row / column sr, k = sk + 2 i): 32 lanes read 512 contiguous bytes
        const int sr = lane & 31, sk = lane >> 5;
        const int src = sr >= rmin ? sr : rmin;
        double (*Wa)[KB_AB_KC][KB_AB_PA] = reinterpret_cast<double (*)[KB_AB_KC][KB_AB_PA]>(&L.stage[wave][0][0][0][0]);    // [re|im][k][row]
        double (*Wb)[KB_AB_KC][KB_AB_PA] = reinterpret_cast<double (*)[KB_AB_KC][KB_AB_PA]>(&L.stage[wave][1][0][0][0]);    // [re|im][k][column]
        const cd* Hblk = H + (size_t)a0 * l + (a0 + kb0s);          // H[kb0s + r, j] of the node = Hblk[r + j * l]
        const unsigned offa0 = (unsigned)src, offb0 = (unsigned)(wave * 32 + sr);
        cd ga[2][4] = {}, gb[2][4] = {};
        auto fetch = [&](int ch, auto SET) {
            constexpr int set = decltype(SET)::value;
            const int chc = ch < nch ? ch : nch - 1;         // (the prefetch runs past the last chunk: stay inside the panel)
            const int jb = n - (chc + 1) * KB_AB_KC;         // rows jb .. jb + 7 (those below k_hi do not exist yet)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int jj = jb + sk + 2 * q;
                const unsigned jc = (unsigned)(jj >= k_hi ? jj : k_hi);
                // (32-bit BYTE offsets from the uniform bases: scalar base + vector offset addressing, two instructions per load)
                ga[set][q] = *reinterpret_cast<const cd*>(reinterpret_cast<const char*>(Hblk) + ((__umul24(jc, (unsigned)l) + offa0) << 4));
                gb[set][q] = *reinterpret_cast<const cd*>(reinterpret_cast<const char*>(P) + (((jc << 7) + offb0) << 4));
            }
        };
        auto stage = [&](int ch, auto SET) {
            constexpr int set = decltype(SET)::value;
            // (branch-free: only the H operand is masked - a zero there annihilates the clamped, finite panel row; a chunk
            // past the last one is staged again and never multiplied)
            const int jb = n - (ch + 1) * KB_AB_KC;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool v = jb + sk + 2 * q >= k_hi;
                Wa[0][sk + 2 * q][sr] = v ? ga[set][q].x : 0.0; Wa[1][sk + 2 * q][sr] = v ? ga[set][q].y : 0.0;
                Wb[0][sk + 2 * q][sr] = gb[set][q].x; Wb[1][sk + 2 * q][sr] = gb[set][q].y;
            }
        };
        double are[2][2] = {}, aim[2][2] = {}, bre[2][2] = {}, bim[2][2] = {};   // [ks half][tile]
        auto operands = [&]() {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    are[h][x] = Wa[0][4 * h + lk][x * 16 + li];
                    aim[h][x] = Wa[1][4 * h + lk][x * 16 + li];
                    bre[h][x] = Wb[0][4 * h + lk][x * 16 + li];
                    bim[h][x] = Wb[1][4 * h + lk][x * 16 + li];
                }
        };
        auto product = [&]() {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        // D[row][col] += A[row][k] B[k][col]: MFMA A operand = H (row = li), B operand = panel (col = li)
                        acc_re[rb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(are[h][rb], bre[h][cb], acc_re[rb][cb], 0, 0, 0);
                        acc_re[rb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(-aim[h][rb], bim[h][cb], acc_re[rb][cb], 0, 0, 0);
                        acc_im[rb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(are[h][rb], bim[h][cb], acc_im[rb][cb], 0, 0, 0);
                        acc_im[rb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(aim[h][rb], bre[h][cb], acc_im[rb][cb], 0, 0, 0);
                    }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        lap(1);
        if (wact) {                                          // (a wavefront whose 16 roots have all settled leaves its SIMD to the
                                                             // other workgroup of the CU: no product, no triangle, no rows)
        fetch(0, S0{});
        fetch(1, S1{});
        stage(0, S0{});
        // (one basic block per chunk; the scheduling hints spread the LDS writes of the next chunk and the global loads of
        // the one after it between the MFMAs instead of in front of them: 8 x {4 MFMA, 2 DS write, 1 VMEM read})
#define KB_AB_INTERLEAVE()                                                   \
        _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) {                   \
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);               \
            __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);               \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);               \
        }
        int ch = 0;
        for (; ch + 1 < nch; ch += 2) {
            // the LDS area holds chunk ch (from set 0), chunk ch + 1 is in flight in set 1
            __builtin_amdgcn_wave_barrier();
            operands();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            stage(ch + 1, S1{});
            fetch(ch + 2, S0{});
            product();
            KB_AB_INTERLEAVE()
            __builtin_amdgcn_wave_barrier();
            operands();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            stage(ch + 2, S0{});
            fetch(ch + 3, S1{});
            product();
            KB_AB_INTERLEAVE()
        }
        if (ch < nch) {                                      // (the number of chunks of a block is odd: 1 + 32 b rows)
            __builtin_amdgcn_wave_barrier();
            operands();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            product();
        }
        }
#undef KB_AB_INTERLEAVE
        lap(2);
#pragma unroll
        for (int i = 0; i < KB_AB_BLK * KB_AB_BLK / 256; ++i) {
            const int e = t + 256 * i, r = e & 31, j = e >> 5;
            L.ht[r][j] = (r >= rmin && j >= rmin && j >= r - 1) ? htr[i] : czero();
        }
        if (t < KB_AB_BLK) L.inv[t] = (kb0s + t >= 1) ? ab_recip(hsub) : mk(-1.0, 0.0);      // k = 0: rho = the sum itself
        __syncthreads();
        lap(3);
        // -- the triangle: every wavefront on its own 32 columns (16 roots: lanes 0-15 x, lanes 16-31 the derivative).  The two
        // halves of the wavefront hold the same columns and share the rows of the block: half h keeps the running sums of the
        // rows r = 2 i + h in REGISTERS.  Step r: the half that owns row r finishes it (one dependent multiply), both halves
        // receive the result (v_permlane32_swap) and add it into their rows above (right-looking), fully unrolled.
        if (wact) {
            const int h = lane >> 5, cl = lane & 31;
            const int c = wave * 32 + cl;                    // this lane's column
            const double isyf = (cl >= 16) ? 1.0 : 0.0;
            const cd z = L.z[wave * 16 + (cl & 15)];
            cd* Tw = reinterpret_cast<cd*>(&L.stage[wave][0][0][0][0]);          // the wavefront's staging area is free now
            // LDS addresses as laundered VGPR bases + compile-time offsets (ds_read_b128 ... offset:imm): left to itself the
            // compiler materialises every wave-uniform address in an SGPR, spills them to VGPR lanes and pays v_readlane + v_mov
            unsigned tww = ab_lds_addr(Tw + lk * KB_AB_TP + li), twl = ab_lds_addr(Tw + h * KB_AB_TP + cl),
                     htb = ab_lds_addr(&L.ht[h][0]), ivb = ab_lds_addr(&L.inv[0]);
            asm volatile("" : "+v"(tww), "+v"(twl), "+v"(htb), "+v"(ivb));
            // accumulators -> one column per lane, 16 rows at a time through the wavefront's area.  D element of lane (li, lk),
            // register g: row = rb*16 + lk + 4 g, column = cb*16 + li.  (The DS operations of a wavefront execute in order.)
            cd rs[KB_AB_BLK / 2];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        ab_lds_st(tww, (4 * g * KB_AB_TP + cb * 16) * (int)sizeof(cd), mk(acc_re[rb][cb][g], acc_im[rb][cb][g]));
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < 8; ++j) rs[rb * 8 + j] = ab_lds_ld(twl, 2 * j * KB_AB_TP * (int)sizeof(cd));
                __builtin_amdgcn_wave_barrier();
            }
            cd pk = L.prow[c];                               // p_k of the step's row (own column)
            if (!(dbg & 2)) {
            // (slots below rmin - rows k < 0 of the topmost block - run through the same code on zeros: basic blocks of eight
            // steps, so that the loads of the next steps' H columns are scheduled under the arithmetic of this one; the topmost
            // block leaves after the group that holds its last row)
#pragma unroll
            for (int r = KB_AB_BLK - 1; r >= 0; --r) {
                if ((r & 7) == 7 && r < KB_AB_BLK - 1 && rmin > r) break;
                constexpr int HALF = KB_AB_BLK / 2;
                const int io = r >> 1, nup = (r + 1) >> 1;   // owner slot; slots updated (the upper half's surplus one at odd r is
                                                             // its own slot io, overwritten below)
                cd hc[HALF];
#pragma unroll
                for (int i = 0; i < HALF; ++i)
                    if (i < nup) hc[i] = ab_lds_ld(htb, (2 * i * (KB_AB_BLK + 1) + (r > 0 ? r - 1 : 0)) * (int)sizeof(cd));   // H[k', k-1], rows 2 i + h
                const cd invr = ab_lds_ld(ivb, r * (int)sizeof(cd));
                // the derivative lane needs x_k: the x lane of the same root is 16 lanes below (v_permlane16_swap)
                const double xkx = ab_row_below(pk.x), xky = ab_row_below(pk.y);
                cd sv = rs[io] - z * pk;
                sv.x = fma(-isyf, xkx, sv.x);
                sv.y = fma(-isyf, xky, sv.y);
                const cd mine = -(sv * invr);                // (row 0 of the matrix, rho itself: inv = -1)
                const cd res = (r & 1) ? mk(ab_upper_half(mine.x), ab_upper_half(mine.y)) : mk(ab_lower_half(mine.x), ab_lower_half(mine.y));
                pk = res;
#pragma unroll
                for (int i = 0; i < HALF; ++i)
                    if (i < nup) cfma(rs[i], hc[i], res);                                   // ... times p_{k-1}
                const bool own = h == (r & 1);
                rs[io] = mk(own ? res.x : rs[io].x, own ? res.y : rs[io].y);
            }
            }
            if (kb0s <= 0) {                                 // the topmost block: row slot rmin holds rho (x lanes) and rho' (y lanes)
                cd rho = rs[0];
#pragma unroll
                for (int i = 1; i < KB_AB_BLK / 2; ++i)
                    if (i == (rmin >> 1)) rho = rs[i];
                if (h == (rmin & 1)) L.prow[c] = rho;
            } else {
                // the finished rows kb0s-1 .. k_hi-1 straight from the registers (half h: the rows 2 i + h; 32 columns are 512
                // contiguous bytes), the row above the next block, and the power-of-two rescaling of a column whose newest x
                // has grown or shrunk a lot (decided on the x lane of the lower half, shared with the derivative lane)
#pragma unroll
                for (int i = 0; i < KB_AB_BLK / 2; ++i)
                    if (!(dbg & 4)) P[(size_t)(kb0s + 2 * i + h - 1) * (2 * KB_AB_TILE) + c] = rs[i];
                const double mx = fmax(fabs(rs[0].x), fabs(rs[0].y));
                int e = 0;
                if (mx > 0.0 && mx == mx && mx < 1.79769313486231570815e308) (void)frexp(mx, &e);
                const double f = ab_row_below((e > 60 || e < -60) ? ldexp(1.0, -e) : 1.0);
                if (h == 0) {
                    L.prow[c] = f * rs[0];
                    L.fac[c] = f;
                    if (f != 1.0) L.flags[1] = 1;
                }
            }
        }
        __syncthreads();
        lap(4);
        if (kb0s <= 0) break;
        if (L.flags[1]) {
            // all finished rows of the rescaled columns (kb0s-1 .. n-1), both x and y
            for (size_t e = t; e < (size_t)(n - (kb0s - 1)) * 2 * KB_AB_TILE; e += 256) {
                const double f = L.fac[e & (2 * KB_AB_TILE - 1)];
                if (f != 1.0) {
                    cd* p = &P[(size_t)(kb0s - 1) * (2 * KB_AB_TILE) + e];
                    *p = f * (*p);
                }
            }
            __syncthreads();
            if (t == 0) L.flags[1] = 0;
            __syncthreads();
        }
        k_hi = kb0s - 1;
        lap(5);
    }
    if (prof) atomicAdd(&abstat[8 * KB_AB_BUDGET + 23], 1);
    // ---- the Aberth update of this tile's roots
    if (t < nr) {
        const int root = t, j = L.ridx[t];
        const cd rho = L.prow[ab_col(root, 0)], rhop = L.prow[ab_col(root, 1)];
        double dz;
        const cd zn = ab_update(L.z[root], rho, rhop, L.S[root], &dz);
        zout[j] = zn;
        ws.lastc[a0 + j] = dz;
        cout[j] = (depth == 0 ? ab_converged(dz, zn, hnorm) : ab_converged_inner(dz, zn, hnorm)) ? 1 : 0;
    }
}

// The tail of a member's root level (kb_aberth.hpp: ab_tail_takes): ONE Aberth iteration of each of its few unsettled
// roots, one wavefront per root, four roots per workgroup.  Hyman's recurrence runs column by column: when x_j is known,
// column j of H (rows 0 .. j) is added to the running row sums, which live in registers (row r on lane r mod 64, chunk
// r / 64: static indices, the loop over the chunk of the pivot row is unrolled); the finished sum of row j comes back
// through v_readlane and gives x_{j-1}.  The derivative runs alongside.  All four wavefronts walk the same columns in the
// same order, so the workgroup streams H through LDS in blocks of KB_AB_TAIL_CB columns (double buffered: the global loads
// of the next block travel under the arithmetic of this one; a step costs no memory latency).  Columns are rescaled by
// powers of two every 32 steps.  grid (KB_AB_TAIL_WGS, members), 256 threads; dynamic LDS: ab_tail_lds_bytes(lmax).
constexpr int KB_AB_TAIL_CB = 4;
constexpr int KB_AB_TAIL_NREG = KB_AB_TAIL_CB * KB_AB_TAIL_MAXL / 256;      // elements of a block per thread
// element i of a thread's share of a block: column i mod CB, row t + 256 (i / CB) - no divisions, coalesced along the rows
#define KB_TAIL_LOAD(B_, regs)                                                                                \
    {                                                                                                      \
        const int rows_ = (CB * (B_) + CB < n) ? CB * (B_) + CB : n;                                       \
        _Pragma("unroll") for (int i = 0; i < KB_AB_TAIL_NREG; ++i) {                                      \
            const int jj_ = i % CB, r_ = t + 256 * (i / CB), j_ = CB * (B_) + jj_;                         \
            regs[i] = (r_ < rows_ && j_ < n) ? H[r_ + (size_t)j_ * l] : kb::czero();                       \
        }                                                                                                  \
    }
#define KB_TAIL_STORE(B_, buf_, regs)                                                                        \
    {                                                                                                      \
        const int rows_ = (CB * (B_) + CB < n) ? CB * (B_) + CB : n;                                       \
        _Pragma("unroll") for (int i = 0; i < KB_AB_TAIL_NREG; ++i) {                                      \
            const int jj_ = i % CB, r_ = t + 256 * (i / CB);                                               \
            if (r_ < rows_) (buf_)[jj_ * npad + r_] = regs[i];                                             \
        }                                                                                                  \
    }
// The steps of the recurrence whose pivot row lies in chunk CJ (rows 64 CJ .. 64 CJ + 63): a template, so that every index
// into the register arrays is a compile-time constant.
template <int CJ>
__device__ __forceinline__ void ab_tail_chunk(kb::cd& sx_, kb::cd& sy_, kb::cd& srho_, kb::cd& srhop_, int& scur_, kb::cd (&Sx)[KB_AB_TAIL_MAXC], kb::cd (&Sy)[KB_AB_TAIL_MAXC],
                                              kb::cd (&regs0)[KB_AB_TAIL_NREG], kb::cd (&regs1)[KB_AB_TAIL_NREG], const kb::cd* __restrict__ H, int l, int n, int npad,
                                              int Bmax, int t, int lane, bool act, kb::cd z, const kb::cd* inv, kb::cd* colbuf) {
    using namespace kb;
    constexpr int MAXC = KB_AB_TAIL_MAXC, CB = KB_AB_TAIL_CB;
    if (CJ * 64 >= n) return;
    const int Bhi = (CJ * (64 / CB) + (64 / CB) - 1 < Bmax) ? CJ * (64 / CB) + (64 / CB) - 1 : Bmax;
    const int Blo = CJ * (64 / CB);                          // (even)
    // One block: two blocks of loads are in flight in the registers (block B - 1 since the last iteration, block B - 2 from now
    // on); RL_ / RS_ = the register set that takes block B - 2 / that holds block B - 1 - fixed by the parity of B, which the
    // loop below keeps static (a set chosen at run time would send both arrays to scratch).
#define KB_TAIL_BODY(B, RL_, RS_)                                                                              \
    {                                                                                                          \
        if ((B) > 1) KB_TAIL_LOAD((B) - 2, RL_)                                                                \
        const cd* buf = colbuf + (size_t)scur_ * CB * npad;                                                    \
        if (act) {                                                                                             \
            _Pragma("unroll") for (int jj = CB - 1; jj >= 0; --jj) {                                           \
                const int j = CB * (B) + jj;                                                                   \
                if (j >= n) continue;                                                                          \
                const int p = j & 63;                                                                          \
                _Pragma("unroll") for (int c = 0; c <= CJ; ++c) {                                              \
                    const int r = lane + 64 * c;                                                               \
                    const cd h = (r <= j) ? buf[jj * npad + r] : czero();                                      \
                    cfma(Sx[c], h, sx_); cfma(Sy[c], h, sy_);                                                  \
                }                                                                                              \
                const cd sj = mk(DevCtx::lane_f64(Sx[CJ].x, p), DevCtx::lane_f64(Sx[CJ].y, p));                \
                const cd spj = mk(DevCtx::lane_f64(Sy[CJ].x, p), DevCtx::lane_f64(Sy[CJ].y, p));               \
                const cd s = sj - z * sx_;                                                                     \
                const cd sp = (spj - z * sy_) - sx_;                                                           \
                if (j > 0) {                                                                                   \
                    const cd iv = inv[j];                                                                      \
                    sx_ = -(s * iv);                                                                           \
                    sy_ = -(sp * iv);                                                                          \
                    if (((n - j) & 31) == 0) {                                                                 \
                        const double mx = fmax(fabs(sx_.x), fabs(sx_.y));                                      \
                        int e = 0;                                                                             \
                        if (mx > 0.0 && mx == mx && mx < 1.79769313486231570815e308) (void)frexp(mx, &e);      \
                        if (e > 60 || e < -60) {                                                               \
                            const double f = ldexp(1.0, -e);                                                   \
                            sx_ = f * sx_; sy_ = f * sy_;                                                      \
                            _Pragma("unroll") for (int c = 0; c < MAXC; ++c) { Sx[c] = f * Sx[c]; Sy[c] = f * Sy[c]; } \
                        }                                                                                      \
                    }                                                                                          \
                } else {                                                                                       \
                    srho_ = s; srhop_ = sp;                                                                    \
                }                                                                                              \
            }                                                                                                  \
        }                                                                                                      \
        if ((B) > 0) KB_TAIL_STORE((B) - 1, colbuf + (size_t)(scur_ ^ 1) * CB * npad, RS_)                     \
        __syncthreads();                                                                                       \
        scur_ ^= 1;                                                                                            \
    }
    // even B: block B - 2 (even) -> regs0, block B - 1 (odd) <- regs1;  odd B: the other way round
    int B = Bhi;
    if (!(B & 1) && B >= Blo) { KB_TAIL_BODY(B, regs0, regs1) --B; }
    for (; B > Blo; B -= 2) {
        KB_TAIL_BODY(B, regs1, regs0)
        KB_TAIL_BODY(B - 1, regs0, regs1)
    }
#undef KB_TAIL_BODY
}
KB_HD int ab_tail_npad(int l) { return (l + 63) & ~63; }
KB_HD int ab_tail_lds_bytes(int l) { return (1 + 2 * KB_AB_TAIL_CB) * ab_tail_npad(l) * (int)sizeof(kb::cd); }
__global__ void __launch_bounds__(256) k_ab_tail(const kb::KbItem* __restrict__ items, const int* __restrict__ perm,
                                                  const kb::cd* __restrict__ arena, const double* __restrict__ varena, double* dcarena,
                                                  const int* __restrict__ needqr, int step, int iter, int* abstat, int npad) {
    using namespace kb;
    constexpr int MAXC = KB_AB_TAIL_MAXC, CB = KB_AB_TAIL_CB;
    const int item = perm[blockIdx.y];
    if (needqr[item]) return;
    const KbItem it = items[item];
    const int l = it.l, depth = ab_depth(l) - 1 - step;
    if (depth != 0 || l > KB_AB_TAIL_MAXL || iter < KB_AB_TAIL_FROM) return;
    const int n = l;
    const AbWs ws = ab_item_ws(it, dcarena);
    // k_ab_iter (launched just before, same iteration) left the number of unsettled roots: most workgroups end here
    if (!ab_tail_takes(l, depth, iter, *ws.nact) || (int)blockIdx.x * 4 >= *ws.nact) return;
    __shared__ int s_cnt[256];
    __shared__ int s_ridx[KB_AB_TAIL_ROOTS];
    __shared__ int s_nact;
    cd* inv = reinterpret_cast<cd*>(kb_smem);                // 1 / H[k, k-1]
    cd* colbuf = inv + npad;                                 // two blocks of CB columns x npad rows
    const int bin = (step * KB_AB_BUDGET + iter) & 1;
    const cd* zin = ws.z[bin];
    cd* zout = ws.z[bin ^ 1];
    const int* cin = ws.conv[bin];
    int* cout = ws.conv[bin ^ 1];
    const cd* __restrict__ H = arena + it.off[KB_BUF_H];
    const double hnorm = varena[it.voff + KB_V_MISC * it.vstride];
    const int t = threadIdx.x;
    // the unsettled roots of the member (the same scan as k_ab_iter's: both see the flags of the previous iteration)
    const int per = (n + 255) / 256;
    const int j0 = t * per, j1 = (j0 + per < n) ? j0 + per : n;
    {
        int c = 0;
        for (int j = j0; j < j1; ++j)
            if (!cin[j]) ++c;
        s_cnt[t] = c;
    }
    __syncthreads();
    int rank = 0;
    for (int u = 0; u < t; ++u) rank += s_cnt[u];
    if (t == 255) s_nact = rank + s_cnt[255];
    __syncthreads();
    const int nactive = s_nact;
    if (!ab_tail_takes(l, depth, iter, nactive)) return;     // k_ab_iter has this member in this iteration
    if (blockIdx.x == 0)
        for (int j = j0; j < j1; ++j)
            if (cin[j]) { zout[j] = zin[j]; cout[j] = 1; }   // settled roots are carried through
    if ((int)blockIdx.x * 4 >= nactive) return;              // no root for this workgroup
    for (int j = j0; j < j1; ++j)
        if (!cin[j]) s_ridx[rank++] = j;
    for (int k = 1 + t; k < n; k += 256) inv[k] = ab_recip(H[k + (size_t)(k - 1) * l]);
    const int wave = t >> 6, lane = t & 63;
    const int q = blockIdx.x * 4 + wave;
    const bool act = q < nactive;                            // (wavefront-uniform)
    // block B = columns CB B .. CB B + CB - 1 (< n), rows 0 .. rows(B) - 1
    const int Bmax = (n - 1) / CB;
    cd regs0[KB_AB_TAIL_NREG], regs1[KB_AB_TAIL_NREG];
    KB_TAIL_LOAD(Bmax, regs0)
    KB_TAIL_STORE(Bmax, colbuf, regs0)
    if (Bmax > 0) { if ((Bmax - 1) & 1) KB_TAIL_LOAD(Bmax - 1, regs1) else KB_TAIL_LOAD(Bmax - 1, regs0) }
    __syncthreads();                                         // (also: s_ridx, inv)
    int me = 0;
    cd z = czero(), S = czero();
    if (act) {
        me = s_ridx[q];
        z = zin[me];
        if (abstat && lane == 0) atomicAdd(&abstat[step * KB_AB_BUDGET + iter], 1);
        // repulsion sum over the other roots (steers the iteration only: reciprocal by the hardware seed + one Newton step)
        double sx = 0.0, sy = 0.0;
        for (int j = lane; j < n; j += 64) {
            if (j == me) continue;
            const cd d = z - zin[j];
            const double den = d.x * d.x + d.y * d.y;
            double iv = __builtin_amdgcn_rcp(den);
            iv = iv * fma(-den, iv, 2.0);
            sx = fma(d.x, iv, sx); sy = fma(-d.y, iv, sy);
        }
        const DevCtx ctx = make_ctx(0);
        S = mk(ctx.wave_sum(sx), ctx.wave_sum(sy));
    }
    // Hyman's recurrence, columns n-1 .. 0
    cd Sx[MAXC], Sy[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) { Sx[c] = czero(); Sy[c] = czero(); }
    cd x = mk(1.0, 0.0), y = czero(), rho = czero(), rhop = czero();
    int cur = 0;
    ab_tail_chunk<7>(x, y, rho, rhop, cur, Sx, Sy, regs0, regs1, H, l, n, npad, Bmax, t, lane, act, z, inv, colbuf);
    ab_tail_chunk<6>(x, y, rho, rhop, cur, Sx, Sy, regs0, regs1, H, l, n, npad, Bmax, t, lane, act, z, inv, colbuf);
    ab_tail_chunk<5>(x, y, rho, rhop, cur, Sx, Sy, regs0, regs1, H, l, n, npad, Bmax, t, lane, act, z, inv, colbuf);
    ab_tail_chunk<4>(x, y, rho, rhop, cur, Sx, Sy, regs0, regs1, H, l, n, npad, Bmax, t, lane, act, z, inv, colbuf);
    ab_tail_chunk<3>(x, y, rho, rhop, cur, Sx, Sy, regs0, regs1, H, l, n, npad, Bmax, t, lane, act, z, inv, colbuf);
    ab_tail_chunk<2>(x, y, rho, rhop, cur, Sx, Sy, regs0, regs1, H, l, n, npad, Bmax, t, lane, act, z, inv, colbuf);
    ab_tail_chunk<1>(x, y, rho, rhop, cur, Sx, Sy, regs0, regs1, H, l, n, npad, Bmax, t, lane, act, z, inv, colbuf);
    ab_tail_chunk<0>(x, y, rho, rhop, cur, Sx, Sy, regs0, regs1, H, l, n, npad, Bmax, t, lane, act, z, inv, colbuf);
    if (act && lane == 0) {
        double dz;
        const cd zn = ab_update(z, rho, rhop, S, &dz);
        zout[me] = zn;
        ws.lastc[me] = dz;
        cout[me] = ab_converged(dz, zn, hnorm) ? 1 : 0;
    }
}

#undef KB_TAIL_LOAD
#undef KB_TAIL_STORE

__global__ void __launch_bounds__(256) k_ab_finish(const kb::KbItem* __restrict__ items, const int* __restrict__ perm, const kb::cd* arena,
                                                    const double* __restrict__ varena, double* dcarena, kb::cd* mu_out, int* needqr) {
    using namespace kb;
    const int item = perm[blockIdx.x];
    if (needqr[item]) return;
    const KbItem it = items[item];
    const DevCtx ctx = make_ctx(0);
    const int l = it.l;
    const AbWs ws = ab_item_ws(it, dcarena);
    const cd* H = arena + it.off[KB_BUF_H];
    const double hnorm = varena[it.voff + KB_V_MISC * it.vstride];
    const int D = ab_depth(l);
    const cd* z = ws.z[(D * KB_AB_BUDGET) & 1];
    int bad = 0;
    cd t1 = czero(), t2 = czero(), s1 = czero(), s2 = czero();
    double a1 = 0.0, a2 = 0.0;
    for (int k = threadIdx.x; k < l; k += blockDim.x) {
        const cd zk = z[k];
        if (!ab_finite(zk) || !ab_acceptable(ws.lastc[k], zk, hnorm)) bad = 1;
        // A small last correction certifies a root only where the iteration converges cubically: the error it leaves is
        // (correction)^3 / (separation)^2.  Inside a cluster tighter than that the convergence is linear and the correction
        // says little about the error (two approximations can sit next to one eigenvalue): such a member goes to the QR
        // iteration (kb_aberth.hpp: ab_certified).
        double sep2 = 1.79769313486231570815e308;
        for (int j = 0; j < l; ++j) {
            const cd d = zk - z[j];
            const double d2 = d.x * d.x + d.y * d.y;
            if (j != k) sep2 = fmin(sep2, d2);
        }
        if (l > 1 && !ab_certified(ws.lastc[k], zk, hnorm, sep2)) bad = 1;
        const cd d = H[k + (size_t)k * l];
        t1 = t1 + d; t2 = t2 + d * d;
        if (k + 1 < l) t2 = t2 + 2.0 * (H[(k + 1) + (size_t)k * l] * H[k + (size_t)(k + 1) * l]);
        s1 = s1 + zk; s2 = s2 + zk * zk;
        a1 += cabs(zk); a2 += abs2(zk);
    }
    t1 = ctx.block_sum(t1); t2 = ctx.block_sum(t2); s1 = ctx.block_sum(s1); s2 = ctx.block_sum(s2);
    a1 = ctx.block_sum(a1); a2 = ctx.block_sum(a2);
    bad = ctx.block_max(bad);
    if (!(cabs(s1 - t1) <= 1e-9 * (a1 + 1e-300)) || !(cabs(s2 - t2) <= 1e-9 * (a2 + 1e-300))) bad = 1;
    if (bad) {
        if (threadIdx.x == 0) needqr[item] = 1;
        return;
    }
    cd* mu = mu_out + it.line_off;
    for (int k = threadIdx.x; k < l; k += blockDim.x) mu[k] = z[k];
}
