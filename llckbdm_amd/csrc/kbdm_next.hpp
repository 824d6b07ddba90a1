// Kernels for the rows next to the hot path: scoring of candidate line lists and cluster
// silhouettes.  Included by kbdm_hip.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "kb_complex.hpp"

namespace kb {

// ------------------------------------------------------------------------------------
// Frequency-domain RMSE of candidate line lists (reference metrics.py:7-17 via min_rmse_kbdm.py:33-38):
//     rmse = sqrt( mean_k ( Re FFT(data)_k / sqrt(N) - Re FFT(est)_k / sqrt(N) )^2 ),
//     est_n = sum_p A_p exp(-t_n / T2_p) exp(i (2 pi F_p t_n + PH_p)),   t_n = n dwell.
// With r = data - est and R = FFT(r):  sum_k (Re R_k)^2 = (1/2) sum_k |R_k|^2 + (1/2) Re sum_k R_k^2
//                                      = (N/2) ( sum_n |r_n|^2 + Re sum_n r_n r_{(N-n) mod N} ),
// so  rmse^2 = ( sum_n |r_n|^2 + Re sum_n r_n r_{(N-n) mod N} ) / (2 N): no transform is needed.
// One workgroup per candidate: the residual goes to a scratch row, then the two sums.
__global__ void __launch_bounds__(256) k_rmse(const cd* __restrict__ data, int N, double dwell,
                                               const double* __restrict__ lines, const long long* __restrict__ cand_off,
                                               cd* __restrict__ resid, double* __restrict__ out) {
    const int c = blockIdx.x;
    const long long p0 = cand_off[c], p1 = cand_off[c + 1];
    cd* r = resid + (size_t)c * N;
    const double twopi = 6.283185307179586476925286766559;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const double t = (double)n * dwell;
        double er = 0.0, ei = 0.0;
        for (long long p = p0; p < p1; ++p) {
            const double A = lines[4 * p], T2 = lines[4 * p + 1], F = lines[4 * p + 2], PH = lines[4 * p + 3];
            const double env = A * exp(-t / T2);
            double sn, cs;
            sincos(twopi * F * t + PH, &sn, &cs);
            er += env * cs;
            ei += env * sn;
        }
        const cd d = data[n];
        r[n] = mk(d.x - er, d.y - ei);
    }
    __syncthreads();
    double s = 0.0;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const cd a = r[n], b = r[(N - n) % N];
        s += a.x * a.x + a.y * a.y + (a.x * b.x - a.y * b.y);
    }
    __shared__ double red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double v = red[0] / (2.0 * (double)N);
        out[c] = sqrt(v > 0.0 ? v : 0.0);
    }
}

// ------------------------------------------------------------------------------------
// Silhouette coefficient of every sample (sklearn.metrics.silhouette_samples semantics, Euclidean
// metric, as used by the reference's clustering sweep llckbdm.py:291): for sample i of class A
//   a = mean distance to the other members of A,  b = min over classes C != A of the mean distance to C,
//   s = (b - a) / max(a, b);  0 for a singleton class (and where max(a, b) = 0).
// The host passes the samples SORTED by class (xs: n x dim row-major, cls[i], class ranges cstart[L+1]),
// so a class is a contiguous range and a thread needs one running sum.  One thread per sample, the
// other samples stream through LDS tiles.  Distances are computed directly (sqrt of the sum of squared
// differences), not through the |x|^2 - 2 x.y + |y|^2 expansion.
#define KB_SIL_TILE 256
#define KB_SIL_MAXDIM 8
__global__ void __launch_bounds__(KB_SIL_TILE) k_silhouette(const double* __restrict__ xs, int n, int dim,
                                                             const int* __restrict__ cls, const int* __restrict__ cstart,
                                                             int nclass, double* __restrict__ out) {
    __shared__ double tile[KB_SIL_TILE * KB_SIL_MAXDIM];
    const int i = blockIdx.x * KB_SIL_TILE + threadIdx.x;
    const bool live = i < n;
    double xi[KB_SIL_MAXDIM];
#pragma unroll
    for (int d = 0; d < KB_SIL_MAXDIM; ++d) xi[d] = (live && d < dim) ? xs[(size_t)i * dim + d] : 0.0;
    const int mine = live ? cls[i] : -1;
    double a = 0.0, b = 1.79769313486231570815e308;
    for (int c = 0; c < nclass; ++c) {
        const int j0 = cstart[c], j1 = cstart[c + 1];
        double sum = 0.0;
        for (int jb = j0; jb < j1; jb += KB_SIL_TILE) {
            const int cnt = (j1 - jb < KB_SIL_TILE) ? j1 - jb : KB_SIL_TILE;
            __syncthreads();
            for (int idx = threadIdx.x; idx < cnt * dim; idx += KB_SIL_TILE) tile[idx] = xs[(size_t)jb * dim + idx];
            __syncthreads();
            if (live) {
                for (int j = 0; j < cnt; ++j) {
                    double d2 = 0.0;
#pragma unroll
                    for (int d = 0; d < KB_SIL_MAXDIM; ++d)
                        if (d < dim) { const double df = xi[d] - tile[j * dim + d]; d2 = fma(df, df, d2); }
                    sum += sqrt(d2);
                }
            }
        }
        const int sz = j1 - j0;
        if (c == mine) a = (sz > 1) ? sum / (double)(sz - 1) : 0.0;
        else if (sz > 0) { const double mean = sum / (double)sz; b = mean < b ? mean : b; }
    }
    if (live) {
        const int sz = cstart[mine + 1] - cstart[mine];
        double s = 0.0;
        if (sz > 1 && nclass > 1) {
            const double mx = a > b ? a : b;
            s = (mx > 0.0) ? (b - a) / mx : 0.0;
        }
        out[i] = s;
    }
}


// The silhouettes of SEVERAL labelings of the same samples in one launch (the clustering sweep scores every fit,
// llckbdm.py:291 inside the loop at :104-110): grid.y = labeling.  The samples stay in their original order in memory
// (one upload); labeling f brings its own stable sort by class - order[f * n + k] = sample at sorted position k - with the
// class of every position and the class ranges (cstart at coff[f], nclass[f] of them).  Thread k takes sorted position k;
// the sums run over the classes and, inside a class, over the sorted positions exactly as in k_silhouette: the same bits
// as one k_silhouette call per labeling.  A labeling with nclass[f] = 0 is skipped (sklearn's precondition fails).
__global__ void __launch_bounds__(KB_SIL_TILE) k_silhouette_sweep(const double* __restrict__ x, int n, int dim,
                                                                   const int* __restrict__ order_all, const int* __restrict__ cls_all,
                                                                   const int* __restrict__ cstart_all, const int* __restrict__ coff,
                                                                   const int* __restrict__ nclass_all, double* __restrict__ out_all) {
    const int f = blockIdx.y;
    const int nclass = nclass_all[f];
    if (nclass == 0) return;
    const int* __restrict__ order = order_all + (size_t)f * n;
    const int* __restrict__ cls = cls_all + (size_t)f * n;
    const int* __restrict__ cstart = cstart_all + coff[f];
    __shared__ double tile[KB_SIL_TILE * KB_SIL_MAXDIM];
    const int i = blockIdx.x * KB_SIL_TILE + threadIdx.x;
    const bool live = i < n;
    const int me = live ? order[i] : 0;
    double xi[KB_SIL_MAXDIM];
#pragma unroll
    for (int d = 0; d < KB_SIL_MAXDIM; ++d) xi[d] = (live && d < dim) ? x[(size_t)me * dim + d] : 0.0;
    const int mine = live ? cls[i] : -1;
    double a = 0.0, b = 1.79769313486231570815e308;
    for (int c = 0; c < nclass; ++c) {
        const int j0 = cstart[c], j1 = cstart[c + 1];
        double sum = 0.0;
        for (int jb = j0; jb < j1; jb += KB_SIL_TILE) {
            const int cnt = (j1 - jb < KB_SIL_TILE) ? j1 - jb : KB_SIL_TILE;
            __syncthreads();
            for (int idx = threadIdx.x; idx < cnt * dim; idx += KB_SIL_TILE) {
                const int j = idx / dim, d = idx - j * dim;
                tile[idx] = x[(size_t)order[jb + j] * dim + d];
            }
            __syncthreads();
            if (live) {
                for (int j = 0; j < cnt; ++j) {
                    double d2 = 0.0;
#pragma unroll
                    for (int d = 0; d < KB_SIL_MAXDIM; ++d)
                        if (d < dim) { const double df = xi[d] - tile[j * dim + d]; d2 = fma(df, df, d2); }
                    sum += sqrt(d2);
                }
            }
        }
        const int sz = j1 - j0;
        if (c == mine) a = (sz > 1) ? sum / (double)(sz - 1) : 0.0;
        else if (sz > 0) { const double mean = sum / (double)sz; b = mean < b ? mean : b; }
    }
    if (live) {
        const int sz = cstart[mine + 1] - cstart[mine];
        double s = 0.0;
        if (sz > 1 && nclass > 1) {
            const double mx = a > b ? a : b;
            s = (mx > 0.0) ? (b - a) / mx : 0.0;
        }
        out_all[(size_t)f * n + me] = s;
    }
}

}  // namespace kb

namespace kb {

// ------------------------------------------------------------------------------------
// HDBSCAN sweep, the O(n^2) parts (the tree part is host code: kbdm_cluster.hpp).
//
// k_knn_dist: for every sample the K smallest distances to the samples (itself included, so entry 0 is 0),
// ascending: knn[i * K + q].  One thread per sample; the running list of a thread lives in LDS laid out
// [q][thread] (conflict-free); candidates stream through registers from an LDS tile; a candidate is inserted only
// if it beats the current last entry of the list (after the first few tiles that is rare).  The workgroup size is
// chosen by the host so that the lists fit the LDS: 64 threads up to K = 300, then 32 / 16 / 8 (K up to ~2300).
// Beyond that the host runs the kernel in PASSES of Kp entries: pass p writes entries koff .. koff + Kp - 1, the
// Kp smallest candidates that come AFTER everything earlier passes wrote, in the order (distance, sample index).
// A pass only needs two numbers per sample from the passes before it: the last distance written (lo) and how many
// candidates of exactly that distance are already out (skipc) - candidates arrive in index order and equal ones keep
// that order in the list, so "skip the first skipc candidates equal to lo" is the continuation.  No limit on K.
#define KB_KNN_TPB 64
__global__ void __launch_bounds__(KB_KNN_TPB) k_knn_dist(const double* __restrict__ xs, int n, int dim, int K,
                                                          double* __restrict__ knn, int koff, int Kp,
                                                          double* __restrict__ lo, int* __restrict__ skipc) {
    extern __shared__ double kb_knn_lds[];
    const int tpb = blockDim.x;
    double* top = kb_knn_lds;                                  // Kp x tpb
    double* tile = kb_knn_lds + (size_t)Kp * tpb;              // tpb x MAXDIM
    const int t = threadIdx.x;
    const int i = blockIdx.x * tpb + t;
    const bool live = i < n;
    double xi[KB_SIL_MAXDIM];
#pragma unroll
    for (int d = 0; d < KB_SIL_MAXDIM; ++d) xi[d] = (live && d < dim) ? xs[(size_t)i * dim + d] : 0.0;
    const double inf = 1.79769313486231570815e308;
    for (int q = 0; q < Kp; ++q) top[q * tpb + t] = inf;
    double kth = inf;                                          // current last entry of this thread's list
    const double lo_d = (koff > 0 && live) ? lo[i] : -1.0;     // distances are >= 0: -1 excludes nothing
    const int skip0 = (koff > 0 && live) ? skipc[i] : 0;
    int skip = skip0;
    for (int jb = 0; jb < n; jb += tpb) {
        const int cnt = (n - jb < tpb) ? n - jb : tpb;
        __syncthreads();
        for (int idx = t; idx < cnt * dim; idx += tpb) tile[idx] = xs[(size_t)jb * dim + idx];
        __syncthreads();
        if (live) {
            for (int j = 0; j < cnt; ++j) {
                double d2 = 0.0;
#pragma unroll
                for (int d = 0; d < KB_SIL_MAXDIM; ++d)
                    if (d < dim) { const double df = xi[d] - tile[j * dim + d]; d2 = fma(df, df, d2); }
                const double dj = sqrt(d2);
                if (dj < kth && dj >= lo_d) {
                    if (dj == lo_d && skip > 0) { --skip; continue; }          // written by an earlier pass
                    // insert into the ascending list behind the entries <= dj (equal entries keep arrival order): the place
                    // by bisection (dj < the last entry), then the tail moves down by one - independent LDS copies, where a
                    // compare-and-shift loop waits for an LDS round trip per entry
                    int lo = 0, hi = Kp - 1;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (top[mid * tpb + t] > dj) hi = mid; else lo = mid + 1;
                    }
                    for (int q = Kp - 1; q > lo; --q) top[q * tpb + t] = top[(q - 1) * tpb + t];
                    top[lo * tpb + t] = dj;
                    kth = top[(Kp - 1) * tpb + t];
                }
            }
        }
    }
    if (live) {
        for (int q = 0; q < Kp; ++q) knn[(size_t)i * K + koff + q] = top[q * tpb + t];
        if (lo) {
            const double last = top[(Kp - 1) * tpb + t];
            int c = 0;
            for (int q = Kp - 1; q >= 0 && top[q * tpb + t] == last; --q) ++c;
            lo[i] = last;
            skipc[i] = (last == lo_d ? skip0 : 0) + c;
        }
    }
}

// k_knn_select: the same table knn[i * K + q] by SELECTION, one workgroup per sample (K <= KB_KNN_SEL_MAXK): the K-th smallest
// distance is found as a 64-bit key (the bit pattern of a non-negative double orders like the double) by radix selection, one
// byte per pass from the top - a pass recomputes the sample's n distances (cheap: the rows of the other samples are a
// contiguous stream out of the L2) and counts, among the keys that match the prefix found so far, the next byte's values (a
// wavefront adds equal bytes with one LDS atomic); then the distances below that key are collected, the list is filled up
// with copies of the key (the ties that straddle position K), sorted (bitonic, in LDS) and written.  The same values in the
// same order as k_knn_dist produces - the distances are computed by the same operations - for a tenth of its time at K = 150:
// k_knn_dist pays an LDS insertion per candidate that beats a thread's list, with the wavefront waiting for its slowest lane.
constexpr int KB_KNN_SEL_MAXK = 4096;
__global__ void __launch_bounds__(256) k_knn_select(const double* __restrict__ xs, int n, int dim, int K, double* __restrict__ knn) {
    const int i = blockIdx.x, t = threadIdx.x, lane = t & 63;
    __shared__ unsigned hist[256];
    __shared__ unsigned long long prefix_s;
    __shared__ int kleft_s, cnt_s;
    extern __shared__ double kb_sel_list[];                     // K entries rounded up to a power of two
    double xi[KB_SIL_MAXDIM];
#pragma unroll
    for (int d = 0; d < KB_SIL_MAXDIM; ++d) xi[d] = (d < dim) ? xs[(size_t)i * dim + d] : 0.0;
    auto dist = [&](int j) {
        double d2 = 0.0;
#pragma unroll
        for (int d = 0; d < KB_SIL_MAXDIM; ++d)
            if (d < dim) { const double df = xi[d] - xs[(size_t)j * dim + d]; d2 = fma(df, df, d2); }
        return sqrt(d2);
    };
    if (t == 0) { prefix_s = 0ull; kleft_s = K; }
    for (int pass = 7; pass >= 0; --pass) {
        hist[t] = 0;
        __syncthreads();
        const unsigned long long prefix = prefix_s;
        const int sh = 8 * pass;
        for (int j0 = 0; j0 < n; j0 += 256) {
            const int j = j0 + t;
            bool valid = false;
            int bin = 0;
            if (j < n) {
                const unsigned long long key = (unsigned long long)__double_as_longlong(dist(j));
                valid = pass == 7 || (key >> (sh + 8)) == (prefix >> (sh + 8));
                bin = (int)((key >> sh) & 255ull);
            }
            // one LDS atomic per distinct byte value and wavefront
            unsigned long long active = __ballot(valid);
            while (active) {
                const int leader = __ffsll((long long)active) - 1;
                const int b = __shfl(bin, leader, 64);
                const unsigned long long same = __ballot(valid && bin == b);
                if (lane == leader) atomicAdd(&hist[b], (unsigned)__popcll(same));
                active &= ~same;
            }
        }
        __syncthreads();
        if (t == 0) {
            int kl = kleft_s, b = 0;
            unsigned c = hist[0];
            while ((int)c < kl) { kl -= (int)c; c = hist[++b]; }      // (the matching keys number at least kl: b stays below 256)
            kleft_s = kl;
            prefix_s = prefix | ((unsigned long long)b << sh);
        }
        __syncthreads();
    }
    const unsigned long long tkey = prefix_s;
    const double tval = __longlong_as_double((long long)tkey);
    int kp2 = 1;
    while (kp2 < K) kp2 <<= 1;
    if (t == 0) cnt_s = 0;
    __syncthreads();
    for (int j = t; j < n; j += 256) {
        const double d = dist(j);
        if ((unsigned long long)__double_as_longlong(d) < tkey) kb_sel_list[atomicAdd(&cnt_s, 1)] = d;      // fewer than K of them
    }
    __syncthreads();
    const int nless = cnt_s;
    for (int q = nless + t; q < kp2; q += 256) kb_sel_list[q] = (q < K) ? tval : 1.79769313486231570815e308;
    __syncthreads();
    for (int size = 2; size <= kp2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int q = t; q < kp2 / 2; q += 256) {
                const int lo = 2 * q - (q & (stride - 1));         // partner pairs (lo, lo + stride)
                const bool up = (lo & size) == 0;
                const double a = kb_sel_list[lo], b = kb_sel_list[lo + stride];
                if ((a > b) == up) { kb_sel_list[lo] = b; kb_sel_list[lo + stride] = a; }
            }
            __syncthreads();
        }
    for (int q = t; q < K; q += 256) knn[(size_t)i * K + q] = kb_sel_list[q];
}

// k_prim_mst: one workgroup per fit (value of min_samples): Prim's algorithm from sample 0 over the complete graph
// with mutual-reachability weights max(core_i, core_j, |x_i - x_j|), core_i = knn[i*K + min_samples - 1].
// Thread t owns samples t, t + nt, ...; per step every thread relaxes its samples against the sample just added
// and proposes its closest outside sample; a workgroup argmin (ties: lowest index) picks the next one.
// scratch per fit: best, core (n doubles each), src (n ints); edges per fit: n - 1 records (a, b, w).
struct KbEdge { int a, b; double w; };
__global__ void __launch_bounds__(1024) k_prim_mst(const double* __restrict__ xs, int n, int dim, int K,
                                                    const double* __restrict__ knn, const int* __restrict__ min_samples,
                                                    double* __restrict__ best_all, double* __restrict__ core_all,
                                                    int* __restrict__ src_all, KbEdge* __restrict__ edges_all) {
    const int fit = blockIdx.x;
    const int ks = min_samples[fit];
    double* best = best_all + (size_t)fit * n;
    double* core = core_all + (size_t)fit * n;                 // contiguous copy of column ks - 1 of knn
    int* src = src_all + (size_t)fit * n;
    KbEdge* edges = edges_all + (size_t)fit * (n - 1);
    const int t = threadIdx.x, nt = blockDim.x;
    const double inf = 1.79769313486231570815e308;
    __shared__ double red_v[16];
    __shared__ int red_i[16];
    __shared__ double xnew[KB_SIL_MAXDIM + 1];
    __shared__ int cur_s;
    for (int i = t; i < n; i += nt) {
        best[i] = (i == 0) ? -1.0 : inf;                       // best < 0: in the tree
        src[i] = 0;
        core[i] = knn[(size_t)i * K + ks - 1];
    }
    if (t == 0) cur_s = 0;
    __syncthreads();
    for (int step = 0; step < n - 1; ++step) {
        const int cur = cur_s;
        if (t <= dim) xnew[t] = (t < dim) ? xs[(size_t)cur * dim + t] : core[cur];
        __syncthreads();
        const double ccur = xnew[dim];
        double mv = inf;
        int mi = 0x7fffffff;
        for (int i = t; i < n; i += nt) {
            double b = best[i];
            if (b >= 0.0) {
                double d2 = 0.0;
#pragma unroll
                for (int d = 0; d < KB_SIL_MAXDIM; ++d)
                    if (d < dim) { const double df = xs[(size_t)i * dim + d] - xnew[d]; d2 = fma(df, df, d2); }
                double w = sqrt(d2);
                w = fmax(w, fmax(core[i], ccur));
                if (w < b) { b = w; best[i] = w; src[i] = cur; }
                if (b < mv) { mv = b; mi = i; }
            }
        }
        // workgroup argmin, ties -> lowest index
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = __shfl_xor(mv, o, 64);
            const int oi = __shfl_xor(mi, o, 64);
            if (ov < mv || (ov == mv && oi < mi)) { mv = ov; mi = oi; }
        }
        if ((t & 63) == 0) { red_v[t >> 6] = mv; red_i[t >> 6] = mi; }
        __syncthreads();
        if (t == 0) {
            double bv = red_v[0];
            int bi = red_i[0];
            for (int wv = 1; wv < (nt >> 6); ++wv)
                if (red_v[wv] < bv || (red_v[wv] == bv && red_i[wv] < bi)) { bv = red_v[wv]; bi = red_i[wv]; }
            edges[step] = KbEdge{src[bi], bi, bv};
            best[bi] = -1.0;
            cur_s = bi;
        }
        __syncthreads();
    }
}


// The same algorithm - the same relaxation, the same argmin, the same edges in the same order - for 4-dimensional samples
// (the reference's transformed line lists) and n <= 1024 NS, with the per-fit state where it is cheap to reach: thread t
// keeps best and core of its samples t, t + 1024, ... in REGISTERS (slot index static: the loops are unrolled), the source
// of a sample's best edge sits in LDS (n ints); per step a thread only streams the coordinates of its samples that are
// still outside the tree (two at a time, four 16-byte loads in flight; a sample inside the tree reads the new vertex's
// row instead: one hot cache line).  Every thread does the final argmin over the wavefronts' candidates itself, the owner
// of the winner records the edge and publishes the new vertex: two barriers per step, no serial tail.
// One step of k_prim_mst costs 16-20 us at n = 20 000 (every access a dependent trip to the L2); this form streams only
// the coordinates: 32 B per outside sample and step.
// (value, index) argmin - the smaller value, the lower index among equal values: a total order, so any reduction tree gives the
// same pair - over the 16 lanes of a DPP row (four cross-lane moves in the vector ALU; the LDS crossbar of __shfl_xor costs
// a round trip per step), then over the wavefront's four rows by v_readlane.  Every lane ends with the wavefront's pair.
template <int CTRL>
__device__ __forceinline__ void kb_argmin_dpp(double& v, int& i) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    const int oi = __builtin_amdgcn_update_dpp(0, i, CTRL, 0xF, 0xF, true);
    const double ov = __hiloint2double(hi, lo);
    if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
}
__device__ __forceinline__ void kb_argmin_wave(double& v, int& i) {
    kb_argmin_dpp<0xB1>(v, i);      // quad_perm [1,0,3,2]
    kb_argmin_dpp<0x4E>(v, i);      // quad_perm [2,3,0,1]
    kb_argmin_dpp<0x141>(v, i);     // row_half_mirror
    kb_argmin_dpp<0x140>(v, i);     // row_mirror: the 16 lanes of a row agree
    double bv = v;
    int bi = i;
#pragma unroll
    for (int l = 0; l < 64; l += 16) {
        const double ov = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
        const int oi = __builtin_amdgcn_readlane(i, l);
        if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    v = bv; i = bi;
}

typedef double kb_prim_d2 __attribute__((ext_vector_type(2)));
// CREG = false (20 480 < n <= 40 000): only the best edges stay in registers, the core distances of the fit are streamed
// with the rows from a contiguous copy (core_all: n doubles per fit).
template <int NS, bool CREG = true>
__global__ void __launch_bounds__(1024) k_prim_mst_reg(const double* __restrict__ xs, int n, int K,
                                                        const double* __restrict__ knn, const int* __restrict__ min_samples,
                                                        double* __restrict__ core_all, KbEdge* __restrict__ edges_all) {
    constexpr int NT = 1024;
    const int fit = blockIdx.x;
    const int ks = min_samples[fit];
    KbEdge* edges = edges_all + (size_t)fit * (n - 1);
    const int t = threadIdx.x;
    const double inf = 1.79769313486231570815e308;
    extern __shared__ int kb_prim_src[];                       // n ints
    __shared__ double red_v[16];
    __shared__ int red_i[16];
    __shared__ double xnew[5];                                 // the new vertex: coordinates, core distance
    __shared__ int cur_s;
    const kb_prim_d2* __restrict__ x2 = reinterpret_cast<const kb_prim_d2*>(xs);
    double* __restrict__ corem = CREG ? nullptr : core_all + (size_t)fit * n;
    double bestr[NS], corer[CREG ? NS : 1];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int i = t + s * NT;
        const double c = (i < n) ? knn[(size_t)i * K + ks - 1] : 0.0;
        if (CREG) corer[s] = c;
        else if (i < n) corem[i] = c;
        bestr[s] = (i < n && i != 0) ? inf : -1.0;            // best < 0: in the tree (or no such sample)
    }
    for (int i = t; i < n; i += NT) kb_prim_src[i] = 0;
    if (t == 0) {
        const kb_prim_d2 a = x2[0], b = x2[1];
        xnew[0] = a.x; xnew[1] = a.y; xnew[2] = b.x; xnew[3] = b.y; xnew[4] = knn[ks - 1];
        cur_s = 0;
    }
    __syncthreads();
#if defined(KB_PANEL_PROF) && defined(__HIP_DEVICE_COMPILE__)
    unsigned long long pt = wall_clock64();
    const bool pon = t == 0 && fit == 0;
#define KB_PRIM_MARK(ph_) do { if (pon) { const unsigned long long q_ = wall_clock64(); atomicAdd(&kb_panel_prof[20 + (ph_)], q_ - pt); pt = q_; } } while (0)
#else
#define KB_PRIM_MARK(ph_) do { } while (0)
#endif
    for (int step = 0; step < n - 1; ++step) {
        const int cur = cur_s;
        const double c0 = xnew[0], c1 = xnew[1], c2 = xnew[2], c3 = xnew[3], ccur = xnew[4];
        double mv = inf;
        int mi = 0x7fffffff;
        // (the thread index goes through an empty asm every step: left alone, the compiler keeps the 64-bit addresses of all
        // NS rows in registers across the step loop - loop invariants that do not fit next to the state)
        int tt = t;
        asm volatile("" : "+v"(tt));
        // rows in flight per batch (the scheduling barrier keeps the next batch's loads - and their registers - behind this
        // batch's arithmetic): two next to the core distances in registers, four when they are streamed
        constexpr int PB = CREG ? 2 : (NS > 30 ? 2 : 4);
#pragma unroll
        for (int s0 = 0; s0 < NS; s0 += PB) {
            kb_prim_d2 xa[PB], xb[PB];
            double cs[PB];
            bool need[PB];
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int s = s0 + u;
                if (s < NS) {
                    const int i = tt + s * NT;
                    // A step is bound by the bytes a compute unit can pull in (641 KB of rows at n = 20 000: 6.5 us), so a
                    // row is requested only by the lanes that need it (masked loads): not for a sample inside the tree, and
                    // not when max(core_i, core_cur) - a lower bound of w - is no better than the sample's best edge
                    need[u] = bestr[s] >= 0.0 && (!CREG || fmax(corer[s], ccur) < bestr[s]);
                    xa[u] = (kb_prim_d2){c0, c1};
                    xb[u] = (kb_prim_d2){c2, c3};
                    cs[u] = CREG ? corer[s] : ccur;
                    if (need[u]) {
                        xa[u] = x2[2 * (size_t)i];
                        xb[u] = x2[2 * (size_t)i + 1];
                        if (!CREG) cs[u] = corem[i];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int s = s0 + u;
                if (s < NS) {
                    const int i = tt + s * NT;
                    double b = bestr[s];
                    const bool act = b >= 0.0;
                    double df = xa[u].x - c0;                  // (the same operations in the same order as k_prim_mst: an fma chain from 0)
                    double d2 = fma(df, df, 0.0);
                    df = xa[u].y - c1; d2 = fma(df, df, d2);
                    df = xb[u].x - c2; d2 = fma(df, df, d2);
                    df = xb[u].y - c3; d2 = fma(df, df, d2);
                    // w >= sqrt(d2) >= b whenever d2 >= b^2 (the square root is monotone and b is a double): the root, the
                    // maxima and the comparison - four fifths of the instructions of a relaxation, and after the first steps
                    // almost never an improvement - run only below fl(b b)(1 + 2^-51), an upper bound of b^2
                    const double b2 = (b * b) * (1.0 + 0x1p-51);
                    if (need[u] && (d2 < b2 || b2 < 1e-290)) {      // (a square that may have underflowed bounds nothing)
                        double w = sqrt(d2);
                        w = fmax(w, fmax(cs[u], ccur));
                        if (w < b) { b = w; bestr[s] = w; kb_prim_src[i] = cur; }
                    }
                    if (act && b < mv) { mv = b; mi = i; }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        KB_PRIM_MARK(0);
        const int mi0 = mi;
        kb_argmin_wave(mv, mi);
        if ((t & 63) == 0) { red_v[t >> 6] = mv; red_i[t >> 6] = mi; }
        KB_PRIM_MARK(1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (LDS traffic only: nobody waits for the edge store's acknowledgement)
        KB_PRIM_MARK(2);
        // the workgroup's pair: every wavefront reduces the 16 candidates itself (lane l < 16 takes wavefront l's)
        double bv = ((t & 63) < NT / 64) ? red_v[t & 63] : inf;
        int bi = ((t & 63) < NT / 64) ? red_i[t & 63] : 0x7fffffff;
        kb_argmin_wave(bv, bi);
        KB_PRIM_MARK(3);
        if (mi0 == bi) {                                       // the owner of the winner (an outside sample exists while step < n - 1)
            edges[step] = KbEdge{kb_prim_src[bi], bi, bv};
            const kb_prim_d2 a = x2[2 * (size_t)bi], b = x2[2 * (size_t)bi + 1];
            double cc = CREG ? 0.0 : corem[bi];
#pragma unroll
            for (int s = 0; s < NS; ++s)
                if (bi == t + s * NT) { if (CREG) cc = corer[s]; bestr[s] = -1.0; }
            xnew[0] = a.x; xnew[1] = a.y; xnew[2] = b.x; xnew[3] = b.y; xnew[4] = cc;
            cur_s = bi;
        }
        KB_PRIM_MARK(4);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        KB_PRIM_MARK(5);
    }
}

}  // namespace kb
