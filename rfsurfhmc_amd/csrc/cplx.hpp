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

// Arithmetic that several kernels must reproduce bit for bit (the layer entries and the vector recurrence of the secular
// functions: a lane per item, 16 lanes per item, producer / consumer blocks) spells out its fused multiply-adds and switches the
// compiler's own contraction off: which of the two products of a * b - c * d the compiler fuses depends on the code around the
// expression (use counts, hoisting), and two kernels built from the same source line then differ in the last bit.
#if defined(__clang__)
#define RFS_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define RFS_NO_CONTRACT
#endif

namespace rfs {

// reciprocal square root: v_rsq_f64 + refinement on the device, 1/sqrt on the host harness
RFS_HD double rsqrt_p(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    // hardware estimate + one coupled Newton / Halley step (error < 1 ulp for normal x); the library's rsqrt() adds
    // a class test and two selects per call for inputs that cannot occur here -- 0 gives inf, NaN gives NaN, as before
    double y = __builtin_amdgcn_rsq(x);
    double e = ::fma(-x * y, y, 1.0);
    return ::fma(y * e, ::fma(e, 0.375, 0.5), y);
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

// ---------------------------------------------------------------------------
// exp and sincos for the argument ranges of this code, written for the f64 VALU: FMA-based Cody-Waite
// reduction + one polynomial each.  The device library's versions carry double-double reductions and
// (for sincos) the Payne-Hanek large-argument path; they cost ~45 / ~120 instructions per call against
// ~20 / ~35 here, and the layer sweeps of every kernel are bounded by exactly those instructions.
// Accuracy: < 1 ulp (exp) and < 1.5 ulp (sin, cos) on the stated ranges (checked against long double libm in the CPU test suite).
// ---------------------------------------------------------------------------
// exp(x) for |x| <= 700.  k = rint(x / ln 2), r = x - k ln2 in two FMA steps (|r| <= 0.3466), Taylor to
// degree 13 (remainder < 4e-18 relative), scaled by 2^k.
// Polynomial coefficients live in constant memory: on the device they arrive in SGPRs through scalar loads and are
// used as the scalar operand of v_fma_f64, instead of costing one v_mov_b64 each on the vector ALU that these
// sweeps saturate (VOP3 has no 64-bit literals on gfx9).
#if defined(__HIP_DEVICE_COMPILE__)
#define RFS_CONST_TABLE __device__ __constant__
#else
#define RFS_CONST_TABLE static const
#endif
RFS_CONST_TABLE double FM_EXP_C[12] = {                   // 1/13! ... 1/2!
    1.6059043836821613e-10, 2.0876756987868100e-09, 2.5052108385441720e-08, 2.7557319223985893e-07,
    2.7557319223985888e-06, 2.4801587301587302e-05, 1.9841269841269841e-04, 1.3888888888888889e-03,
    8.3333333333333332e-03, 4.1666666666666664e-02, 1.6666666666666666e-01, 0.5};
RFS_CONST_TABLE double FM_SIN_C[6] = {                    // fdlibm S6 ... S1
    1.58969099521155010221e-10, -2.50507602534068634195e-08, 2.75573137070700676789e-06,
    -1.98412698298579493134e-04, 8.33333333332248946124e-03, -1.66666666666666324348e-01};
RFS_CONST_TABLE double FM_COS_C[6] = {                    // fdlibm C6 ... C1
    -1.13596475577881948265e-11, 2.08757232129817482790e-09, -2.75573143513906633035e-07,
    2.48015872894767294178e-05, -1.38888888888741095749e-03, 4.16666666666666019037e-02};
RFS_CONST_TABLE double FM_RED_C[6] = {                    // log2(e), ln2 hi, ln2 lo | 2/pi, pi/2 hi, mid  (lo below)
    1.4426950408889634074, 6.93147180369123816490e-01, 1.90821492927058770002e-10,
    6.36619772367581382433e-01, 1.57079632679489655800e+00, 6.12323399573676603587e-17};

RFS_HD double fm_exp(double x) {
    const double k = rint(x * FM_RED_C[0]);
    double r = ::fma(-k, FM_RED_C[1], x);                     // ln2 high part (fdlibm split)
    r = ::fma(-k, FM_RED_C[2], r);                            // ln2 low part
    double p = FM_EXP_C[0];
#pragma unroll
    for (int i = 1; i < 12; i++) p = ::fma(p, r, FM_EXP_C[i]);
    p = ::fma(p, r, 1.0);
    p = ::fma(p, r, 1.0);
    return ldexp(p, (int)k);
}

// sin and cos of x for |x| < 1e9 (phases here are layer thickness x vertical wavenumber: at most a few 1e3; no
// large-argument path is carried).  n = rint(x 2/pi), r = x - n pi/2 with pi/2 split in three parts (each step
// one FMA; the split is good to 2^-160, so the reduction error stays below 1e-30 n), fdlibm's kernel polynomials
// on |r| <= pi/4, quadrant fix-up by n mod 4.  NaN / inf in -> NaN out.
RFS_HD void fm_sincos(double x, double* sn, double* cs) {
    const double n = rint(x * FM_RED_C[3]);
    double r = ::fma(-n, FM_RED_C[4], x);                     // pi/2 rounded to double
    r = ::fma(-n, FM_RED_C[5], r);                            // next 53 bits
    r = ::fma(-n, -1.49738490485916983294e-33, r);            // and the next
    const double z = r * r;
    double ps = FM_SIN_C[0], pc = FM_COS_C[0];
#pragma unroll
    for (int i = 1; i < 6; i++) { ps = ::fma(ps, z, FM_SIN_C[i]); pc = ::fma(pc, z, FM_COS_C[i]); }
    const double s0 = ::fma(ps * z, r, r);                    // r + r^3 (S1 + ...)
    const double c0 = ::fma(pc * z, z, ::fma(-0.5, z, 1.0));  // 1 - z/2 + z^2 (C1 + ...)
    const int q = (int)n & 3;
    const double ss = (q & 1) ? c0 : s0, cc = (q & 1) ? s0 : c0;
    *sn = (q & 2) ? -ss : ss;
    *cs = ((q + 1) & 2) ? -cc : cc;
}

// The same sine / cosine with the twelve polynomial coefficients handed in by the caller, who holds them in VECTOR registers
// (fm_vc_load, once per kernel): a kernel whose layer loop keeps more 64-bit constants alive than the scalar register file
// takes (k_swd_exact: two exponentials and two sine / cosine pairs side by side) otherwise re-reads the spilled ones with
// v_readlane in every layer -- 24 of 361 vector instructions.  Same operations in the same order: same numbers bit for bit.
struct FmVC { double s[6], c[6]; };
RFS_HD FmVC fm_vc_load() {
    FmVC k;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        k.s[i] = FM_SIN_C[i]; k.c[i] = FM_COS_C[i];
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(k.s[i]));      // (opaque to the compiler from here on: stays in vector registers)
        asm volatile("" : "+v"(k.c[i]));
#endif
    }
    return k;
}
RFS_HD void fm_sincos_vc(double x, const FmVC& k, double* sn, double* cs) {
    const double n = rint(x * FM_RED_C[3]);
    double r = ::fma(-n, FM_RED_C[4], x);
    r = ::fma(-n, FM_RED_C[5], r);
    r = ::fma(-n, -1.49738490485916983294e-33, r);
    const double z = r * r;
    double ps = k.s[0], pc = k.c[0];
#pragma unroll
    for (int i = 1; i < 6; i++) { ps = ::fma(ps, z, k.s[i]); pc = ::fma(pc, z, k.c[i]); }
    const double s0 = ::fma(ps * z, r, r);
    const double c0 = ::fma(pc * z, z, ::fma(-0.5, z, 1.0));
    const int q = (int)n & 3;
    const double ss = (q & 1) ? c0 : s0, cc = (q & 1) ? s0 : c0;
    *sn = (q & 2) ? -ss : ss;
    *cs = ((q + 1) & 2) ? -cc : cc;
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
RFS_HD cplx inv(cplx a) { double d = rcp_p(norm2(a)); return cplx{a.re * d, -a.im * d}; }
RFS_HD cplx operator/(cplx a, cplx b) { return a * inv(b); }
RFS_HD cplx operator/(cplx a, double s) { double d = rcp_p(s); return cplx{a.re * d, a.im * d}; }
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
    double e = fm_exp(z.re), s, c;
    fm_sincos(z.im, &s, &c);
    return cplx{e * c, e * s};
}

}  // namespace rfs
