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
    // order this wavefront's LDS/global writes before later reads by its other lanes
    __device__ __forceinline__ void wave_fence() const {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ char* scratch() const { return smem + KB_RED_BYTES; }
    __device__ __forceinline__ int scratch_bytes() const { return smem_bytes - KB_RED_BYTES; }

    __device__ __forceinline__ double wave_sum(double v) const {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    }
    __device__ __forceinline__ cd wave_sum(cd v) const {
        return mk(wave_sum(v.x), wave_sum(v.y));
    }
    __device__ __forceinline__ double wave_max(double v) const {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
        return v;
    }
    __device__ __forceinline__ int wave_max(int v) const {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            int t = __shfl_xor(v, o, 64);
            v = t > v ? t : v;
        }
        return v;
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
    void wave_fence() const {}
    char* scratch() const { return smem + KB_RED_BYTES; }
    int scratch_bytes() const { return smem_bytes - KB_RED_BYTES; }
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
