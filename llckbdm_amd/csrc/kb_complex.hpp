// Complex-double scalar helpers shared by every KBDM kernel.
//
// Plain struct instead of hip/thrust complex so that the same header compiles as
// gfx950 device code (hipcc) and inside the host simulation used by the CPU unit
// tests (tests/hostsim, g++).  Everything is FP64: the reference path is complex128
// end to end (reference kbdm.py:111-113) and U0 has cond ~1e16, so no reduced
// precision anywhere.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define KB_HD __host__ __device__ __forceinline__
#else
#define KB_HD inline
#endif

namespace kb {

struct alignas(16) cd {
    double x, y;
};

KB_HD cd mk(double x, double y) { cd r; r.x = x; r.y = y; return r; }
KB_HD cd czero() { return mk(0.0, 0.0); }
KB_HD cd operator+(cd a, cd b) { return mk(a.x + b.x, a.y + b.y); }
KB_HD cd operator-(cd a, cd b) { return mk(a.x - b.x, a.y - b.y); }
KB_HD cd operator-(cd a) { return mk(-a.x, -a.y); }
KB_HD cd operator*(cd a, cd b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
KB_HD cd operator*(double s, cd a) { return mk(s * a.x, s * a.y); }
KB_HD cd operator*(cd a, double s) { return mk(s * a.x, s * a.y); }
KB_HD cd& operator+=(cd& a, cd b) { a.x += b.x; a.y += b.y; return a; }
KB_HD cd& operator-=(cd& a, cd b) { a.x -= b.x; a.y -= b.y; return a; }
KB_HD cd conj(cd a) { return mk(a.x, -a.y); }
// conj(a) * b
KB_HD cd cmulc(cd a, cd b) { return mk(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x); }
// acc += a*b as four FMAs
KB_HD void cfma(cd& acc, cd a, cd b) {
    acc.x = fma(a.x, b.x, acc.x);
    acc.x = fma(-a.y, b.y, acc.x);
    acc.y = fma(a.x, b.y, acc.y);
    acc.y = fma(a.y, b.x, acc.y);
}
// acc += conj(a)*b
KB_HD void cfmac(cd& acc, cd a, cd b) {
    acc.x = fma(a.x, b.x, acc.x);
    acc.x = fma(a.y, b.y, acc.x);
    acc.y = fma(a.x, b.y, acc.y);
    acc.y = fma(-a.y, b.x, acc.y);
}
KB_HD double abs2(cd a) { return a.x * a.x + a.y * a.y; }
KB_HD double cabs(cd a) { return hypot(a.x, a.y); }
KB_HD double cabs1(cd a) { return fabs(a.x) + fabs(a.y); }
KB_HD bool is_zero(cd a) { return a.x == 0.0 && a.y == 0.0; }

// a / b, Smith's algorithm (robust against over/underflow of |b|^2, like LAPACK zladiv)
KB_HD cd cdiv(cd a, cd b) {
    if (fabs(b.x) >= fabs(b.y)) {
        double r = b.y / b.x;
        double den = b.x + b.y * r;
        return mk((a.x + a.y * r) / den, (a.y - a.x * r) / den);
    } else {
        double r = b.x / b.y;
        double den = b.y + b.x * r;
        return mk((a.x * r + a.y) / den, (a.y * r - a.x) / den);
    }
}

// principal square root
KB_HD cd csqrt_(cd z) {
    double a = cabs(z);
    if (a == 0.0) return czero();
    double re = sqrt(0.5 * (a + fabs(z.x)));
    double im = z.y / (2.0 * re);
    if (z.x >= 0.0) return mk(re, im);
    // re, im swap roles on the left half plane
    return mk(fabs(im), (z.y >= 0.0) ? re : -re);
}

constexpr double KB_EPS = 1.1102230246251565e-16;    // 2^-53  (LAPACK dlamch('E'))
constexpr double KB_ULP = 2.2204460492503131e-16;    // 2^-52  (LAPACK dlamch('P'))
constexpr double KB_SAFMIN = 2.2250738585072014e-308;

}  // namespace kb
