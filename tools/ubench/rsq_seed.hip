// How accurate are the v_rsq_f64 / v_rcp_f64 hardware seeds on gfx950, and what is left after one and two
// Newton steps?  Prints max relative errors over 2^20 random arguments spanning 1e-300..1e300.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>

__global__ void k(const double* x, double* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double t = x[i];
    double rs = __builtin_amdgcn_rsq(t);
    out[6 * i + 0] = rs;
    rs = rs * fma(-0.5 * t * rs, rs, 1.5);
    out[6 * i + 1] = rs;
    rs = rs * fma(-0.5 * t * rs, rs, 1.5);
    out[6 * i + 2] = rs;
    double r = __builtin_amdgcn_rcp(t);
    out[6 * i + 3] = r;
    r = r * fma(-t, r, 2.0);
    out[6 * i + 4] = r;
    r = r * fma(-t, r, 2.0);
    out[6 * i + 5] = r;
}

int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), o(6 * n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> m(1.0, 2.0), e(-300.0, 300.0);
    for (int i = 0; i < n; ++i) x[i] = m(g) * pow(10.0, (i < n / 2) ? e(g) : e(g) * 0.01);
    double *dx, *dout;
    hipMalloc(&dx, 8 * n); hipMalloc(&dout, 48 * n);
    hipMemcpy(dx, x.data(), 8 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(o.data(), dout, 48 * n, hipMemcpyDeviceToHost);
    double err[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const long double t = x[i];
        const long double rs = 1.0L / sqrtl(t), r = 1.0L / t;
        for (int q = 0; q < 3; ++q) err[q] = fmax(err[q], (double)fabsl((o[6 * i + q] - rs) / rs));
        for (int q = 3; q < 6; ++q) err[q] = fmax(err[q], (double)fabsl((o[6 * i + q] - r) / r));
    }
    printf("v_rsq_f64 seed %.3e (2^%.1f)  +1 Newton %.3e  +2 Newton %.3e\n", err[0], log2(err[0]), err[1], err[2]);
    printf("v_rcp_f64 seed %.3e (2^%.1f)  +1 Newton %.3e  +2 Newton %.3e\n", err[3], log2(err[3]), err[4], err[5]);
    return 0;
}
