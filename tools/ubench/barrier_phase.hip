// Microbenchmark: cost of one "barrier phase" on gfx950 (LDS read -> short FP64 chain -> LDS write -> s_barrier)
// as a function of the number of wavefronts in the workgroup.  Build: hipcc -O3 --offload-arch=gfx950 -o barrier_phase barrier_phase.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(1024) k_phase(long long* out, int iters, double* sink) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < 8192; i += nt) lds[i] = 1.0 + 1e-9 * i;
    __syncthreads();
    double acc = 0.0;
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 1) {
            const int idx = (tid * 7 + it) & 8191;
            double a = lds[idx], b = lds[(idx + 57) & 8191];
            if (MODE >= 2) {
#pragma unroll
                for (int k = 0; k < 8; ++k) a = fma(a, 0.999999, b);
            }
            lds[idx] = a;
            acc += a;
        }
        __syncthreads();
    }
    const long long c1 = clock64();
    if (tid == 0) out[blockIdx.x] = c1 - c0;
    if (acc == 12345.678) sink[0] = acc;
}

int main() {
    long long* d_out; double* d_sink;
    hipMalloc(&d_out, 8 * 16); hipMalloc(&d_sink, 8);
    const int iters = 20000;
    for (int mode = 0; mode < 3; ++mode)
        for (int nw : {1, 2, 4, 8, 16}) {
            auto fn = mode == 0 ? k_phase<0> : (mode == 1 ? k_phase<1> : k_phase<2>);
            hipLaunchKernelGGL(fn, dim3(1), dim3(64 * nw), 65536, 0, d_out, iters, d_sink);
            hipDeviceSynchronize();
            long long h = 0;
            hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
            printf("mode %d (0=barrier only,1=+lds rw,2=+8 dep fma) waves %2d : %.1f clk/phase\n", mode, nw, (double)h / iters);
        }
    return 0;
}
