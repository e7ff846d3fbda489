// Minimal f64 complex arithmetic for gfx950 device code (HIP has no native complex
// transcendental; everything is built from real sincos/exp/sqrt).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RFS_HD __host__ __device__ __forceinline__
#else
#define RFS_HD inline
#endif

namespace rfs {

// reciprocal square root: v_rsq_f64 + refinement on the device, 1/sqrt on the host harness
RFS_HD double rsqrt_p(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return rsqrt(x);
#else
    return 1.0 / sqrt(x);
#endif
}

// reciprocal to ~1 ulp: hardware estimate + two Newton steps on the device (shorter dependent
// chain than the IEEE division expansion); plain division on the host harness
RFS_HD double rcp_p(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
    r = ::fma(::fma(-x, r, 1.0), r, r);
    r = ::fma(::fma(-x, r, 1.0), r, r);
    return r;
#else
    return 1.0 / x;
#endif
}

struct cplx {
    double re, im;
};

RFS_HD cplx C(double re, double im = 0.0) { return cplx{re, im}; }
RFS_HD cplx operator+(cplx a, cplx b) { return cplx{a.re + b.re, a.im + b.im}; }
RFS_HD cplx operator-(cplx a, cplx b) { return cplx{a.re - b.re, a.im - b.im}; }
RFS_HD cplx operator-(cplx a) { return cplx{-a.re, -a.im}; }
RFS_HD cplx operator*(cplx a, cplx b) { return cplx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
RFS_HD cplx operator*(double s, cplx a) { return cplx{s * a.re, s * a.im}; }
RFS_HD cplx operator*(cplx a, double s) { return cplx{s * a.re, s * a.im}; }
RFS_HD cplx operator+(cplx a, double s) { return cplx{a.re + s, a.im}; }
RFS_HD cplx operator+(double s, cplx a) { return cplx{a.re + s, a.im}; }
RFS_HD cplx operator-(cplx a, double s) { return cplx{a.re - s, a.im}; }
RFS_HD cplx operator-(double s, cplx a) { return cplx{s - a.re, -a.im}; }
RFS_HD cplx& operator+=(cplx& a, cplx b) { a.re += b.re; a.im += b.im; return a; }
RFS_HD cplx conj(cplx a) { return cplx{a.re, -a.im}; }
RFS_HD cplx mul_i(cplx a) { return cplx{-a.im, a.re}; }       // i*a
RFS_HD double norm2(cplx a) { return a.re * a.re + a.im * a.im; }
RFS_HD cplx inv(cplx a) { double d = 1.0 / norm2(a); return cplx{a.re * d, -a.im * d}; }
RFS_HD cplx operator/(cplx a, cplx b) { return a * inv(b); }
RFS_HD cplx operator/(cplx a, double s) { double d = 1.0 / s; return cplx{a.re * d, a.im * d}; }
RFS_HD cplx operator/(double s, cplx a) { return s * inv(a); }
// a*b + c
RFS_HD cplx fma(cplx a, cplx b, cplx c) {
    return cplx{::fma(a.re, b.re, ::fma(-a.im, b.im, c.re)), ::fma(a.re, b.im, ::fma(a.im, b.re, c.im))};
}
// Re(a*b)
RFS_HD double re_mul(cplx a, cplx b) { return a.re * b.re - a.im * b.im; }

// principal square root (same branch as C csqrt / Fortran sqrt for finite non-axis input)
RFS_HD cplx csqrt_p(cplx z) {
    double m = sqrt(z.re * z.re + z.im * z.im);
    double t = sqrt(0.5 * (m + fabs(z.re)));
    if (t == 0.0) return cplx{0.0, 0.0};
    double u = 0.5 * z.im / t;
    if (z.re >= 0.0) return cplx{t, u};
    return cplx{fabs(u), copysign(t, z.im)};
}
// exp(z)
RFS_HD cplx cexp_p(cplx z) {
    double e = exp(z.re), s, c;
    sincos(z.im, &s, &c);
    return cplx{e * c, e * s};
}

}  // namespace rfs
