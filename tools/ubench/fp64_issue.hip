// Microbenchmark (gfx950): what a one-CU-per-member FP64 kernel pays per instruction.
//   dep chain / independent chains of v_fma_f64, v_rsq_f64, v_rcp_f64, ds_read_b128 round trip, v_readlane, s_barrier,
// as a function of wavefronts per workgroup (1 workgroup on the chip: 4 waves = one per SIMD, 8 = two per SIMD).
// Build: hipcc -O3 --offload-arch=gfx950 -o fp64_issue fp64_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(1024) k(long long* out, int iters, double* sink, double seed) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += blockDim.x) lds[i] = 1.0 + 1e-9 * i;
    __syncthreads();
    double a0 = seed + tid * 1e-9, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double c = 0.9999999, d = 1e-7;
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {            // 16 dependent FMAs
#pragma unroll
            for (int k = 0; k < 16; ++k) a0 = fma(a0, c, d);
        } else if (MODE == 1) {     // 8 independent chains x 2 = 16 FMAs
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                a0 = fma(a0, c, d); a1 = fma(a1, c, d); a2 = fma(a2, c, d); a3 = fma(a3, c, d);
                a4 = fma(a4, c, d); a5 = fma(a5, c, d); a6 = fma(a6, c, d); a7 = fma(a7, c, d);
            }
        } else if (MODE == 2) {     // 4 dependent rsq
#pragma unroll
            for (int k = 0; k < 4; ++k) a0 = __builtin_amdgcn_rsq(a0) + 1.0;
        } else if (MODE == 3) {     // 4 dependent rcp
#pragma unroll
            for (int k = 0; k < 4; ++k) a0 = __builtin_amdgcn_rcp(a0) + 1.0;
        } else if (MODE == 4) {     // 4 dependent ds_read_b64 round trips (address from the value read)
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int idx = ((int)a0 + tid) & 4095; a0 = lds[idx]; }
        } else if (MODE == 5) {     // 16 readlane + use
#pragma unroll
            for (int k = 0; k < 16; ++k) a0 += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a0), (k * 3) & 63), __builtin_amdgcn_readlane(__double2loint(a0), (k * 5) & 63));
        } else if (MODE == 6) {     // barrier only
            __syncthreads();
        } else if (MODE == 7) {     // 2 independent chains x 8
#pragma unroll
            for (int k = 0; k < 8; ++k) { a0 = fma(a0, c, d); a1 = fma(a1, c, d); }
        } else if (MODE == 8) {     // 4 independent chains x 4
#pragma unroll
            for (int k = 0; k < 4; ++k) { a0 = fma(a0, c, d); a1 = fma(a1, c, d); a2 = fma(a2, c, d); a3 = fma(a3, c, d); }
        } else if (MODE == 9) {     // 16 dependent FP32 FMAs (reference)
            float f = (float)a0;
#pragma unroll
            for (int k = 0; k < 16; ++k) f = fmaf(f, 0.99999f, 1e-6f);
            a0 = f;
        } else if (MODE == 10) {    // lds write + barrier + lds read (one hand-off)
            lds[tid] = a0;
            __syncthreads();
            a0 = lds[(tid + 64) & (blockDim.x - 1)];
        }
    }
    const long long c1 = clock64();
    if (tid == 0) out[blockIdx.x] = c1 - c0;
    const double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 12345.678) sink[0] = s;
}

int main() {
    long long* d_out; double* d_sink;
    hipMalloc(&d_out, 8 * 16); hipMalloc(&d_sink, 8);
    const int iters = 200000;
    const char* names[] = {"16 dependent v_fma_f64", "16 v_fma_f64 in 8 chains", "4 dependent (v_rsq_f64 + add)", "4 dependent (v_rcp_f64 + add)",
                           "4 dependent ds_read_b64 (+cvt)", "16 x (2 v_readlane + add)", "s_barrier", "16 v_fma_f64 in 2 chains",
                           "16 v_fma_f64 in 4 chains", "16 dependent v_fma_f32", "ds_write + barrier + ds_read"};
    void (*fns[])(long long*, int, double*, double) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>};
    for (int mode = 0; mode < 11; ++mode)
        for (int nw : {1, 4, 8, 16}) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(fns[mode], dim3(1), dim3(64 * nw), 65536, 0, d_out, iters, d_sink, 1.5);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            long long h = 0;
            hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
            printf("%-34s waves %2d : %8.1f ticks/iter  %8.1f ns/iter  (%.2f ticks/ns)\n", names[mode], nw, (double)h / iters,
                   ms * 1e6 / iters, (double)h / (ms * 1e6));
        }
    return 0;
}
