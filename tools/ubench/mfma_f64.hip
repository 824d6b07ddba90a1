// v_mfma_f64_16x16x4_f64 issue rate on gfx950: independent accumulators, 1 / 2 / 4 wavefronts per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(double* out, long long* ticks, int iters) {
    d4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    __syncthreads();
    const long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    __syncthreads();
    const long long t1 = wall_clock64();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
    double* out; long long* ticks;
    hipMalloc(&out, sizeof(double) * 1024 * 1024); hipMalloc(&ticks, sizeof(long long) * 1024);
    const int iters = 20000;
    for (int threads : {64, 256, 512, 1024}) {
        for (int blocks : {1, 256}) {
            hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, ticks, iters);
            hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, ticks, iters); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h; hipMemcpy(&h, ticks, sizeof(h), hipMemcpyDeviceToHost);
            const double n_mfma = 8.0 * iters;                     // per wavefront
            const double waves = threads / 64.0 * blocks;
            printf("threads %4d blocks %3d: %.1f ns per MFMA per wavefront (10 ns ticks: %.1f), %.2f TFLOP/s total\n", threads, blocks,
                   ms * 1e6 / n_mfma, h * 10.0 / n_mfma, waves * n_mfma * 2048.0 / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
