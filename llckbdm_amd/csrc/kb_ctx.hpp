// Execution contexts for the per-ensemble-member algorithms in kb_core.hpp.
//
// The factorisation routines are written once against a small "workgroup" interface
// (thread / lane / wave ids, barrier, wave and block reductions, a scratch arena).
//   * DevCtx  : one gfx950 workgroup; 64-lane wavefronts, LDS scratch, s_barrier.
//   * HostCtx : a one-thread "workgroup" (wave size 1) used ONLY by tests/hostsim to
//               debug index/convergence logic on the CPU build box, which has no GPU.
//               It is never compiled into, or loaded by, the product library.
#pragma once
#include "kb_complex.hpp"

namespace kb {

// Scratch layout: the first KB_RED_BYTES of the arena are reserved for block reductions.
constexpr int KB_RED_SLOTS = 64;                                   // >= max waves per block
constexpr int KB_RED_BYTES = KB_RED_SLOTS * 2 * (int)sizeof(double);

#if defined(__HIPCC__)
struct DevCtx {
    static constexpr int WS = 64;
    char* smem;        // dynamic LDS base, 16-byte aligned
    int smem_bytes;    // total bytes available (including the reduction slots)

    __device__ __forceinline__ int tid() const { return threadIdx.x; }
    __device__ __forceinline__ int nthreads() const { return blockDim.x; }
    __device__ __forceinline__ int lane() const { return threadIdx.x & 63; }
    __device__ __forceinline__ int wave() const { return threadIdx.x >> 6; }
    __device__ __forceinline__ int nwaves() const { return blockDim.x >> 6; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    // Barrier for data exchanged through LDS only: waits for this wavefront's LDS operations, NOT for its outstanding global
    // stores (__syncthreads waits for their acknowledgement: 1-2 us when a phase has just stored a vector to memory).
    // Global memory written before it is NOT ordered for the other wavefronts.
    __device__ __forceinline__ void sync_lds() const {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // order this wavefront's LDS/global writes before later reads by its other lanes
    __device__ __forceinline__ void wave_fence() const {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    // order this wavefront's LDS writes before later LDS reads by its other lanes WITHOUT waiting for its
    // outstanding global stores (wave_fence waits for their acknowledgement: hundreds of cycles per use)
    __device__ __forceinline__ void lds_fence() const {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    // Ordering only: the DS instructions of ONE wavefront execute in issue order, so a read issued after a write - from
    // whatever lanes - returns the written data without waiting for the write's completion (lgkmcnt); what must not
    // happen is the compiler moving LDS accesses across this point.
    __device__ __forceinline__ void lds_order() const {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ char* scratch() const { return smem + KB_RED_BYTES; }
    __device__ __forceinline__ int scratch_bytes() const { return smem_bytes - KB_RED_BYTES; }

    // Wavefront reductions without the LDS crossbar: four DPP butterfly stages inside each row of 16
    // lanes (quad_perm 1, quad_perm 2, row_half_mirror, row_mirror), then the four row results through
    // v_readlane.  Every stage combines the same two partial values in every lane (commutative ops), so
    // all 64 lanes end with identical bits.  All lanes of the wavefront must be active.
    template <int CTRL>
    __device__ static __forceinline__ double dpp_f64(double v) {
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
        return __hiloint2double(hi, lo);
    }
    __device__ static __forceinline__ double lane_f64(double v, int l) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                                __builtin_amdgcn_readlane(__double2loint(v), l));
    }
    // sum over each group of four adjacent lanes (identical bits in the four); all lanes of the wavefront must be active
    static constexpr int QUAD = 4;
    __device__ __forceinline__ double quad_sum(double v) const {
        v += dpp_f64<0xB1>(v);      // quad_perm [1,0,3,2]
        v += dpp_f64<0x4E>(v);      // quad_perm [2,3,0,1]
        return v;
    }
    // sum over each group of eight adjacent lanes (identical bits in the eight): ((l0+l1)+(l2+l3)) + ((l4+l5)+(l6+l7))
    static constexpr int OCT = 8;
    __device__ __forceinline__ double oct_sum(double v) const {
        v += dpp_f64<0xB1>(v);      // quad_perm [1,0,3,2]
        v += dpp_f64<0x4E>(v);      // quad_perm [2,3,0,1]
        v += dpp_f64<0x141>(v);     // row_half_mirror
        return v;
    }
    __device__ __forceinline__ cd oct_sum(cd v) const { return mk(oct_sum(v.x), oct_sum(v.y)); }
    __device__ __forceinline__ double wave_sum(double v) const {
        v += dpp_f64<0xB1>(v);      // quad_perm [1,0,3,2]
        v += dpp_f64<0x4E>(v);      // quad_perm [2,3,0,1]
        v += dpp_f64<0x141>(v);     // row_half_mirror
        v += dpp_f64<0x140>(v);     // row_mirror
        return (lane_f64(v, 0) + lane_f64(v, 16)) + (lane_f64(v, 32) + lane_f64(v, 48));
    }
    __device__ __forceinline__ cd wave_sum(cd v) const {
        return mk(wave_sum(v.x), wave_sum(v.y));
    }
    __device__ __forceinline__ double wave_max(double v) const {
        v = fmax(v, dpp_f64<0xB1>(v));
        v = fmax(v, dpp_f64<0x4E>(v));
        v = fmax(v, dpp_f64<0x141>(v));
        v = fmax(v, dpp_f64<0x140>(v));
        return fmax(fmax(lane_f64(v, 0), lane_f64(v, 16)), fmax(lane_f64(v, 32), lane_f64(v, 48)));
    }
    __device__ __forceinline__ int wave_max(int v) const {
        int t;
        t = __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true); v = t > v ? t : v;
        t = __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true); v = t > v ? t : v;
        t = __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true); v = t > v ? t : v;
        t = __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true); v = t > v ? t : v;
        const int a0 = __builtin_amdgcn_readlane(v, 0), a1 = __builtin_amdgcn_readlane(v, 16);
        const int a2 = __builtin_amdgcn_readlane(v, 32), a3 = __builtin_amdgcn_readlane(v, 48);
        const int m0 = a0 > a1 ? a0 : a1, m1 = a2 > a3 ? a2 : a3;
        return m0 > m1 ? m0 : m1;
    }
    // Block reductions: every thread must call; every thread receives the same bits
    // (fixed summation order), so data-dependent branches taken on the result are uniform.
    __device__ __forceinline__ double block_sum(double v) const {
        double* red = reinterpret_cast<double*>(smem);
        v = wave_sum(v);
        if (lane() == 0) red[wave()] = v;
        __syncthreads();
        double t = 0.0;
        const int nw = nwaves();
        for (int w = 0; w < nw; ++w) t += red[w];
        __syncthreads();
        return t;
    }
    // block_sum_l in two halves with the caller's barrier (sync_lds) between them: work that only needs the barrier, not the
    // sum, goes in front of it.  The slots are free again after the NEXT barrier behind red_get.
    __device__ __forceinline__ void red_put(double v) const {
        double* red = reinterpret_cast<double*>(smem);
        v = wave_sum(v);
        if (lane() == 0) red[wave()] = v;
    }
    __device__ __forceinline__ double red_get() const {
        const double* red = reinterpret_cast<const double*>(smem);
        double t = 0.0;
        const int nw = nwaves();
        for (int w = 0; w < nw; ++w) t += red[w];
        return t;
    }
    // the same with LDS-only barriers (sync_lds)
    __device__ __forceinline__ double block_sum_l(double v) const {
        double* red = reinterpret_cast<double*>(smem);
        v = wave_sum(v);
        if (lane() == 0) red[wave()] = v;
        sync_lds();
        double t = 0.0;
        const int nw = nwaves();
        for (int w = 0; w < nw; ++w) t += red[w];
        sync_lds();
        return t;
    }
    __device__ __forceinline__ double block_max_l(double v) const {
        double* red = reinterpret_cast<double*>(smem);
        v = wave_max(v);
        if (lane() == 0) red[wave()] = v;
        sync_lds();
        double t = red[0];
        const int nw = nwaves();
        for (int w = 1; w < nw; ++w) t = fmax(t, red[w]);
        sync_lds();
        return t;
    }
    __device__ __forceinline__ cd block_sum(cd v) const {
        double* red = reinterpret_cast<double*>(smem);
        v = wave_sum(v);
        if (lane() == 0) { red[2 * wave()] = v.x; red[2 * wave() + 1] = v.y; }
        __syncthreads();
        cd t = czero();
        const int nw = nwaves();
        for (int w = 0; w < nw; ++w) { t.x += red[2 * w]; t.y += red[2 * w + 1]; }
        __syncthreads();
        return t;
    }
    __device__ __forceinline__ double block_max(double v) const {
        double* red = reinterpret_cast<double*>(smem);
        v = wave_max(v);
        if (lane() == 0) red[wave()] = v;
        __syncthreads();
        double t = red[0];
        const int nw = nwaves();
        for (int w = 1; w < nw; ++w) t = fmax(t, red[w]);
        __syncthreads();
        return t;
    }
    __device__ __forceinline__ int block_max(int v) const {
        int* red = reinterpret_cast<int*>(smem);
        v = wave_max(v);
        if (lane() == 0) red[wave()] = v;
        __syncthreads();
        int t = red[0];
        const int nw = nwaves();
        for (int w = 1; w < nw; ++w) t = red[w] > t ? red[w] : t;
        __syncthreads();
        return t;
    }
};
#endif  // __HIPCC__

#if !defined(__HIP_DEVICE_COMPILE__)
// One-thread workgroup for the CPU debugging build (tests/hostsim only).
struct HostCtx {
    static constexpr int WS = 1;
    char* smem;
    int smem_bytes;
    int tid() const { return 0; }
    int nthreads() const { return 1; }
    int lane() const { return 0; }
    int wave() const { return 0; }
    int nwaves() const { return 1; }
    void sync() const {}
    void sync_lds() const {}
    static constexpr int OCT = 1;
    double oct_sum(double v) const { return v; }
    cd oct_sum(cd v) const { return v; }
    double block_sum_l(double v) const { return v; }
    double block_max_l(double v) const { return v; }
    void red_put(double v) const { *reinterpret_cast<double*>(smem) = v; }
    double red_get() const { return *reinterpret_cast<const double*>(smem); }
    void wave_fence() const {}
    void lds_fence() const {}
    void lds_order() const {}
    char* scratch() const { return smem + KB_RED_BYTES; }
    int scratch_bytes() const { return smem_bytes - KB_RED_BYTES; }
    static constexpr int QUAD = 1;
    double quad_sum(double v) const { return v; }
    double wave_sum(double v) const { return v; }
    cd wave_sum(cd v) const { return v; }
    double wave_max(double v) const { return v; }
    int wave_max(int v) const { return v; }
    double block_sum(double v) const { return v; }
    cd block_sum(cd v) const { return v; }
    double block_max(double v) const { return v; }
    int block_max(int v) const { return v; }
};
#endif

}  // namespace kb
