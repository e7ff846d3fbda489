// Receiver-function propagator math for one (chain, frequency) lane.
//
// Re-design of the reference's frequency-domain RF core (RFModule.f90:592-987):
// instead of rebuilding the full 4x4 product once per (parameter, layer) pair
// -- O(4 n^2) matrix products per frequency, cal_response_par_all :670-705 -- a
// lane carries
//   pass A (bottom-up): the single ROW  r_j = e_row^T E^-1 A_{n-1} ... A_{j+1}
//   pass B (top-down) : the single COLUMN y_j = A_{j-1} ... A_1 (u e_c1 + v e_c2)
// so that  sum_k Re(u dR21/dm_j + v dR22'/dm_j) = Re( r_j . dA_j/dm . y_j ),
// i.e. O(n) work and O(1) state per lane.  (u, v) are the per-frequency adjoint
// weights produced from the FFT of the weighted residual (rf_kernels.hip), which
// replaces the reference's 4n inverse FFTs by one forward FFT.
//
// Everything that does not depend on frequency (RFModule.f90:721-736: va_k, vb_k,
// gamma, gamma1, mu, ...) lives in RfLayer, computed once per (chain, layer).
#pragma once
#include "cplx.hpp"

namespace rfs {

// rf_type: 1 = P (row 2: R21 = M(2,1), R22 = i M(2,2)), 2 = S (row 1: R21 = M(1,2), R22 = -i M(1,1))
// RFModule.f90:653-658.

// ---- float32 complex numbers of the sweep beyond the band (rf_row_step_f32 below) ----
#if defined(__clang__) && !defined(RFS_NO_PK)
// (re, im) as one two-lane vector: sums, differences and the two halves of a complex product map onto the packed f32
// instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32, half selection and negation folded into their modifiers)
typedef float rfs_f32x2 __attribute__((ext_vector_type(2)));
struct cplxf { rfs_f32x2 v; };
RFS_HD cplxf cf(float re, float im) { cplxf o; o.v = rfs_f32x2{re, im}; return o; }
RFS_HD float cf_re(cplxf a) { return a.v.x; }
RFS_HD float cf_im(cplxf a) { return a.v.y; }
RFS_HD cplxf operator+(cplxf a, cplxf b) { cplxf o; o.v = a.v + b.v; return o; }
RFS_HD cplxf operator-(cplxf a, cplxf b) { cplxf o; o.v = a.v - b.v; return o; }
RFS_HD cplxf operator-(cplxf a) { cplxf o; o.v = -a.v; return o; }
RFS_HD cplxf operator*(cplxf a, cplxf b) {
    cplxf o;
    const rfs_f32x2 bs = __builtin_shufflevector(b.v, -b.v, 3, 0);          // (-b.im, b.re)
    o.v = a.v.xx * b.v + a.v.yy * bs;
    return o;
}
RFS_HD cplxf operator*(float s, cplxf a) { cplxf o; o.v = a.v * s; return o; }
RFS_HD cplxf cmadd(cplxf acc, cplxf a, cplxf b) {      // acc + a b: two packed FMAs
    cplxf o;
    const rfs_f32x2 bs = __builtin_shufflevector(b.v, -b.v, 3, 0);
    o.v = a.v.yy * bs + (a.v.xx * b.v + acc.v);
    return o;
}
RFS_HD cplxf cmsub(cplxf acc, cplxf a, cplxf b) {      // acc - a b
    cplxf o;
    const rfs_f32x2 bs = __builtin_shufflevector(b.v, -b.v, 3, 0);
    o.v = (acc.v - a.v.xx * b.v) - a.v.yy * bs;
    return o;
}
RFS_HD cplxf cf_pmul(float a0, float a1, float b0, float b1) { cplxf o; o.v = rfs_f32x2{a0, a1} * rfs_f32x2{b0, b1}; return o; }   // (a0 b0, a1 b1)
#else
struct cplxf { float re, im; };
RFS_HD cplxf cf(float re, float im) { return cplxf{re, im}; }
RFS_HD float cf_re(cplxf a) { return a.re; }
RFS_HD float cf_im(cplxf a) { return a.im; }
RFS_HD cplxf operator+(cplxf a, cplxf b) { return cplxf{a.re + b.re, a.im + b.im}; }
RFS_HD cplxf operator-(cplxf a, cplxf b) { return cplxf{a.re - b.re, a.im - b.im}; }
RFS_HD cplxf operator-(cplxf a) { return cplxf{-a.re, -a.im}; }
RFS_HD cplxf operator*(cplxf a, cplxf b) { return cplxf{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
RFS_HD cplxf operator*(float s, cplxf a) { return cplxf{s * a.re, s * a.im}; }
RFS_HD cplxf cmadd(cplxf acc, cplxf a, cplxf b) { return acc + a * b; }
RFS_HD cplxf cmsub(cplxf acc, cplxf a, cplxf b) { return acc - a * b; }
RFS_HD cplxf cf_pmul(float a0, float a1, float b0, float b1) { return cplxf{a0 * b0, a1 * b1}; }
#endif
RFS_HD cplxf to_f32(cplx a) { return cf((float)a.re, (float)a.im); }

struct RfLayer {          // frequency-independent constants of one finite layer / the half-space
    cplx pva, pvb;        // sqrt(p^2 - 1/alpha^2), sqrt(p^2 - 1/beta^2)  (= p*va_k, p*vb_k)
    cplx va, vb, iva, ivb;// va_k, vb_k and reciprocals
    cplx iva2;            // 1/va_k^2
    cplx mu2, imu2;       // 2*mu, 1/(2*mu)
    cplx gam, gam1, gam2, gam3;
    cplx ia, ib;          // 1/alpha, 1/beta
    cplx sca, scb;        // alpha/vp, beta/vs  (RFModule.f90:624-628)
    double h, rho;
    // float32 copies for rf_row_step_f32: va, vb, 1/va, 1/vb, gam1, gam1^2, 1/(2 mu), 2 mu, 2 mu gam1, gam
    cplxf fva, fvb, fiva, fivb, fg1, fg1sq, fim2, fm2, fm2g1, fgam;
};
constexpr int RF_LAYER_DOUBLES = sizeof(RfLayer) / sizeof(double);

RFS_HD void rf_make_layer(RfLayer& L, double h, double rho, double vp, double vs, double qa,
                          double qb, double p) {
    // complex velocities, RFModule.f90:377-378
    cplx alpha = vp * C(1.0 + 1.0 / (8.0 * qa * qa), 1.0 / (2.0 * qa));
    cplx beta = vs * C(1.0 + 1.0 / (8.0 * qb * qb), 1.0 / (2.0 * qb));
    cplx ia = inv(alpha), ib = inv(beta);
    cplx b2 = beta * beta;
    L.pva = csqrt_p(p * p - ia * ia);
    L.pvb = csqrt_p(p * p - ib * ib);
    L.va = L.pva / p; L.vb = L.pvb / p;
    L.iva = inv(L.va); L.ivb = inv(L.vb);
    L.iva2 = L.iva * L.iva;
    cplx mu = rho * b2;
    L.mu2 = 2.0 * mu; L.imu2 = inv(L.mu2);
    L.gam = (2.0 * p * p) * b2;
    L.gam1 = 1.0 - inv(L.gam);
    cplx ap = alpha * p;
    L.gam2 = L.gam * inv(ap * ap);
    L.gam3 = inv(L.gam - 2.0);
    L.ia = ia; L.ib = ib;
    L.sca = alpha / vp; L.scb = beta / vs;
    L.h = h; L.rho = rho;
    L.fva = to_f32(L.va); L.fvb = to_f32(L.vb); L.fiva = to_f32(L.iva); L.fivb = to_f32(L.ivb);
    L.fg1 = to_f32(L.gam1); L.fg1sq = to_f32(L.gam1 * L.gam1); L.fim2 = to_f32(L.imu2); L.fm2 = to_f32(L.mu2);
    L.fm2g1 = to_f32(L.mu2 * L.gam1); L.fgam = to_f32(L.gam);
}

struct RfHyp {            // per (layer, frequency): cosh and the four scaled sinh terms
    cplx ca, cb, xa, ya, xb, yb;
    double sa, sb;        // branch sign: v_alpha = sa * omega * pva (principal sqrt, :729)
};

// nu*h = s*omega*pva*h with s chosen so that nu is the principal square root.
// In two halves: the transcendental part -- e^{Re}, cos / sin of the imaginary part for the P and the S leg, six numbers -- and
// everything that follows from it.  The row sweep (pass A) can leave the six numbers of every (layer, band frequency) in HBM
// for the column sweep (pass B), which then skips two exponentials and two sine / cosine pairs per layer ("rf_store_hyp").
struct RfHypB { double e1, c1, s1, e2, c2, s2; };
RFS_HD void rf_hyp_base(const RfLayer& L, cplx omega, RfHypB& B) {
    cplx ta = omega * L.pva, tb = omega * L.pvb;
    B.e1 = fm_exp(L.h * ta.re); B.e2 = fm_exp(L.h * tb.re);
    fm_sincos(L.h * ta.im, &B.s1, &B.c1);
    fm_sincos(L.h * tb.im, &B.s2, &B.c2);
}
RFS_HD void rf_hyp_from(const RfLayer& L, cplx omega, const RfHypB& B, RfHyp& H) {
    cplx ta = omega * L.pva, tb = omega * L.pvb;
    H.sa = (ta.re > 0.0 || (ta.re == 0.0 && ta.im >= 0.0)) ? 1.0 : -1.0;
    H.sb = (tb.re > 0.0 || (tb.re == 0.0 && tb.im >= 0.0)) ? 1.0 : -1.0;
    // cosh / sinh of the complex arguments from ONE exp, one reciprocal and one sincos each:
    // e^{a+ib} = e^a (c + i s), e^{-(a+ib)} = e^{-a} (c - i s)  ->  cosh = (ch c, sh s), sinh = (sh c, ch s)
    const double e1 = B.e1, s1 = B.s1, c1 = B.c1, e2 = B.e2, s2 = B.s2, c2 = B.c2;
    double i1 = rcp_p(e1), i2 = rcp_p(e2);
    double ch1 = 0.5 * (e1 + i1), sh1 = 0.5 * (e1 - i1), ch2 = 0.5 * (e2 + i2), sh2 = 0.5 * (e2 - i2);
    H.ca = cplx{ch1 * c1, sh1 * s1}; H.cb = cplx{ch2 * c2, sh2 * s2};
    cplx sha = H.sa * cplx{sh1 * c1, ch1 * s1}, shb = H.sb * cplx{sh2 * c2, ch2 * s2};
    H.xa = L.va * sha; H.ya = sha * L.iva;
    H.xb = L.vb * shb; H.yb = shb * L.ivb;
}
RFS_HD void rf_hyp(const RfLayer& L, cplx omega, RfHyp& H) {
    RfHypB B;
    rf_hyp_base(L, omega, B);
    rf_hyp_from(L, omega, B, H);
}

struct V4 { cplx v[4]; };

// Haskell layer matrix entries (RFModule.f90:746-763), 11 distinct values, WITHOUT their common factor gamma:
// the products below apply it once to the resulting vector (4 complex multiplications instead of 11).
struct RfA { cplx a11, a12, a13, a14, a21, a22, a23, a24, a31, a32, a41, g; };

RFS_HD void rf_build_A(const RfLayer& L, const RfHyp& H, RfA& A) {
    cplx g1 = L.gam1;
    cplx dc = H.ca - H.cb;
    cplx dci = dc * L.imu2, g1sq = g1 * g1;
    A.g = L.gam;
    A.a11 = H.ca - g1 * H.cb;
    A.a12 = g1 * H.ya - H.xb;
    A.a13 = -dci;
    A.a14 = (H.xb - H.ya) * L.imu2;
    A.a21 = g1 * H.yb - H.xa;
    A.a22 = H.cb - g1 * H.ca;
    A.a23 = (H.xa - H.yb) * L.imu2;
    A.a24 = dci;
    A.a31 = L.mu2 * (g1 * dc);
    A.a32 = L.mu2 * (g1sq * H.ya - H.xb);
    A.a41 = L.mu2 * (g1sq * H.yb - H.xa);
    // a33 = a22, a34 = -a12, a42 = -a31, a43 = -a21, a44 = a11
}

RFS_HD V4 rf_row_times_A(const V4& r, const RfA& A) {   // r' = r . A
    V4 o;
    o.v[0] = A.g * (r.v[0] * A.a11 + r.v[1] * A.a21 + r.v[2] * A.a31 + r.v[3] * A.a41);
    o.v[1] = A.g * (r.v[0] * A.a12 + r.v[1] * A.a22 + r.v[2] * A.a32 - r.v[3] * A.a31);
    o.v[2] = A.g * (r.v[0] * A.a13 + r.v[1] * A.a23 + r.v[2] * A.a22 - r.v[3] * A.a21);
    o.v[3] = A.g * (r.v[0] * A.a14 + r.v[1] * A.a24 - r.v[2] * A.a12 + r.v[3] * A.a11);
    return o;
}

// How much the layer matrices of a chain can grow: sum over the finite layers of h (sigma (|Im p_alpha| + |Im p_beta|) + w_max (|Re p_alpha| +
// |Re p_beta|)) -- the exponent of exp(+- nu h) at the band's highest frequency (nu = omega p_v, omega = w - i sigma: the
// damping acts on the propagating S leg, the larger one; a post-critical leg is evanescent in full).  Peeling a row off with A^-1 loses exp(2 x this) of relative accuracy, so a chain peels only below
// RF_PEEL_EMAX (1e-16 e^{10} = 2e-12; a soak of 600 random configurations: gradients within 4e-11 of the stored-row sweep at 6); deeper / slower stacks, shorter windows (sigma = 4 / window) or post-critical
// slownesses keep their stored rows.  Decided per chain, on the device, by both sweeps from the same numbers.
constexpr double RF_PEEL_EMAX = 5.0;
RFS_HD double rf_growth_exponent(const RfLayer* L, int n, double sigma, double wmax) {
    double e = 0.0;
    for (int j = 0; j < n - 1; j++)
        e += L[j].h * (sigma * (fabs(L[j].pvb.im) + fabs(L[j].pva.im)) + wmax * (fabs(L[j].pva.re) + fabs(L[j].pvb.re)));
    return e;
}

// r' = r . A^-1.  A is the layer's propagator over its thickness h (the factor gamma does not depend on h), so its inverse
// is the propagator over -h: the cosh-type entries stay, the sinh-type ones (a12, a14, a21, a23, a32, a41) change sign.
// Pass B peels the layers off pass A's FINAL row with it instead of reading a stored row per layer -- sound where the
// waves propagate inside the layers (teleseismic slownesses: |exp(nu h)| stays near 1 and nothing is amplified).
RFS_HD V4 rf_row_times_Ainv(const V4& r, const RfA& A) {
    V4 o;
    o.v[0] = A.g * (r.v[0] * A.a11 - r.v[1] * A.a21 + r.v[2] * A.a31 - r.v[3] * A.a41);
    o.v[1] = A.g * (r.v[1] * A.a22 - r.v[0] * A.a12 - r.v[2] * A.a32 - r.v[3] * A.a31);
    o.v[2] = A.g * (r.v[0] * A.a13 - r.v[1] * A.a23 + r.v[2] * A.a22 + r.v[3] * A.a21);
    o.v[3] = A.g * (r.v[1] * A.a24 - r.v[0] * A.a14 + r.v[2] * A.a12 + r.v[3] * A.a11);
    return o;
}

RFS_HD V4 rf_A_times_col(const RfA& A, const V4& y) {   // y' = A . y
    V4 o;
    o.v[0] = A.g * (A.a11 * y.v[0] + A.a12 * y.v[1] + A.a13 * y.v[2] + A.a14 * y.v[3]);
    o.v[1] = A.g * (A.a21 * y.v[0] + A.a22 * y.v[1] + A.a23 * y.v[2] + A.a24 * y.v[3]);
    o.v[2] = A.g * (A.a31 * y.v[0] + A.a32 * y.v[1] + A.a22 * y.v[2] - A.a12 * y.v[3]);
    o.v[3] = A.g * (A.a41 * y.v[0] - A.a31 * y.v[1] - A.a21 * y.v[2] + A.a11 * y.v[3]);
    return o;
}

// ---------------------------------------------------------------------------------------------------------------
// float32 form of one pass-A step r' = r . A_j, for the frequencies BEYOND the Gaussian band.  Those frequencies reach
// the results only through the water level (a maximum of |R21|^2 over all frequencies, RFModule.f90:396-398) and through
// spectrum values weighted by exp(-(w/2f0)^2) < 1e-11: what is needed from them is an upper bound of their maximum
// (k_rf_mid1 decides from it, exactly, whether their exact values can matter at all, and recomputes them in f64 where
// they can).  Phases are reduced in f64 (a layer holds up to ~100 rad of them), everything else runs on the f32 VALU
// with the hardware exp2 / sin / cos.  Relative error of |R21|^2 for n layers of growth exponent E: ~ n eps32 e^{2E}
// (measured 4e-6 at 30 layers, E = 1.5), see RF_F32_BOUND.
// ---------------------------------------------------------------------------------------------------------------
struct V4f { cplxf v[4]; };

RFS_HD float f32_exp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_exp2f(x * 1.44269504f);
#else
    return expf(x);
#endif
}
RFS_HD void f32_sincos_rev(float x, float* s, float* c) {     // sin / cos of 2 pi x, x in [0, 1)
#if defined(__HIP_DEVICE_COMPILE__)
    *s = __builtin_amdgcn_sinf(x); *c = __builtin_amdgcn_cosf(x);
#else
    *s = sinf(6.28318531f * x); *c = cosf(6.28318531f * x);
#endif
}
RFS_HD float f32_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}

// the f32 sweep is used for a chain only while  RF_F32_BOUND n e^{2E} <= RF_F32_MARGIN  (E: rf_growth_exponent at the
// Nyquist frequency); k_rf_mid1 then takes (1 + RF_F32_MARGIN) x the f32 maximum as the bound of the true one
constexpr double RF_F32_MARGIN = 1.0e-2, RF_F32_BOUND = 16.0 * 6.0e-8;
RFS_HD double rf_f32_emax(int n) { return 0.5 * log(RF_F32_MARGIN / (RF_F32_BOUND * (n > 1 ? n : 1))); }

// k_rf_mid1's decision for a chain whose frequencies beyond the band were swept in float32.  b1 / b2: maxima of |R21|^2 and
// |R21^2|^2 over the band (exact), lo1 / lo2: their minima over the band (exact), h1 / h2: the maxima over the other
// frequencies as the float32 sweep gave them (true value <= (1 + RF_F32_MARGIN) x that, squared for |R21^2|^2).
//   0: the band holds both maxima -- b1, b2 ARE the maxima over all frequencies, the water level is the all-f64 one
//   1: no band frequency can reach a water level set by the bounds: whatever the true maxima are, fai = max(|R21|^2,
//      water max) picks |R21|^2 at every band frequency (RFModule.f90:396-401), and so does the adjoint's fai2
//   2: neither: the frequencies beyond the band are swept again in f64
RFS_HD int rf_f32_decide(double water, double b1, double b2, double h1, double h2, double lo1, double lo2) {
    const double u1 = h1 * (1.0 + RF_F32_MARGIN), u2 = h2 * ((1.0 + RF_F32_MARGIN) * (1.0 + RF_F32_MARGIN));
    if (u1 <= b1 && u2 <= b2) return 0;
    if (lo1 >= water * fmax(b1, u1) && lo2 >= water * fmax(b2, u2)) return 1;
    return 2;
}

RFS_HD V4f rf_row_step_f32(const RfLayer& L, cplx omega, const V4f& r) {
    const cplx ta = omega * L.pva, tb = omega * L.pvb;
    const float sa = (ta.re > 0.0 || (ta.re == 0.0 && ta.im >= 0.0)) ? 1.0f : -1.0f;
    const float sb = (tb.re > 0.0 || (tb.re == 0.0 && tb.im >= 0.0)) ? 1.0f : -1.0f;
    double pa = (L.h * ta.im) * 0.15915494309189535, pb = (L.h * tb.im) * 0.15915494309189535;
    pa -= floor(pa); pb -= floor(pb);
    float s1, c1, s2, c2;
    f32_sincos_rev((float)pa, &s1, &c1); f32_sincos_rev((float)pb, &s2, &c2);
    const float e1 = f32_exp((float)(L.h * ta.re)), e2 = f32_exp((float)(L.h * tb.re));
    const float i1 = f32_rcp(e1), i2 = f32_rcp(e2);
    const float ch1 = 0.5f * (e1 + i1), sh1 = 0.5f * (e1 - i1), ch2 = 0.5f * (e2 + i2), sh2 = 0.5f * (e2 - i2);
    const cplxf ca = cf_pmul(ch1, sh1, c1, s1), cb = cf_pmul(ch2, sh2, c2, s2);
    const cplxf sha = sa * cf_pmul(sh1, ch1, c1, s1), shb = sb * cf_pmul(sh2, ch2, c2, s2);
    const cplxf xa = L.fva * sha, ya = sha * L.fiva, xb = L.fvb * shb, yb = shb * L.fivb;
    const cplxf g1 = L.fg1, g1sq = L.fg1sq, im2 = L.fim2, m2 = L.fm2;
    const cplxf dc = ca - cb, dci = dc * im2;
    const cplxf a11 = cmsub(ca, g1, cb), a22 = cmsub(cb, g1, ca);
    const cplxf a12 = cmadd(-xb, g1, ya), a21 = cmadd(-xa, g1, yb);
    const cplxf a14 = (xb - ya) * im2, a23 = (xa - yb) * im2;
    const cplxf a31 = L.fm2g1 * dc, a32 = m2 * cmadd(-xb, g1sq, ya), a41 = m2 * cmadd(-xa, g1sq, yb);
    // a13 = -dci, a24 = dci; a33 = a22, a34 = -a12, a42 = -a31, a43 = -a21, a44 = a11
    V4f o;
    o.v[0] = L.fgam * cmadd(cmadd(cmadd(r.v[0] * a11, r.v[1], a21), r.v[2], a31), r.v[3], a41);
    o.v[1] = L.fgam * cmsub(cmadd(cmadd(r.v[0] * a12, r.v[1], a22), r.v[2], a32), r.v[3], a31);
    o.v[2] = L.fgam * cmsub(cmadd(cmsub(r.v[1] * a23, r.v[0], dci), r.v[2], a22), r.v[3], a21);
    o.v[3] = L.fgam * cmadd(cmsub(cmadd(r.v[0] * a14, r.v[1], dci), r.v[2], a12), r.v[3], a11);
    return o;
}

// Row `rf_type` of the half-space matrix E^-1 (RFModule.f90:903-920).
RFS_HD V4 rf_einv_row(const RfLayer& L, int rf_type) {
    V4 r;
    cplx hg = 0.5 * L.gam;
    if (rf_type == 1) {
        r.v[0] = hg * (L.gam1 * L.ivb);
        r.v[1] = hg;
        r.v[2] = hg * (-(L.imu2 * L.ivb));
        r.v[3] = hg * (-L.imu2);
    } else {
        r.v[0] = -hg;
        r.v[1] = hg * (-(L.gam1 * L.iva));
        r.v[2] = hg * L.imu2;
        r.v[3] = hg * (L.imu2 * L.iva);
    }
    return r;
}

// r . (dA/dm) . y for the four parameter classes of one finite layer, in the
// reference's order [rho, vp, vs, thk] (RFModule.f90:771, 811-874), including the
// complex-velocity rescaling alpha/vp, beta/vs (:624-628).  k = omega * p.
// RHO = false leaves T[0] to the caller (pass B with row peeling has both neighbouring rows and forms it from the
// commutator, rf_rho_partial).
template <bool RHO = true>
RFS_HD void rf_layer_partials(const RfLayer& L, const RfHyp& H, cplx k, const V4& r,
                              const V4& y, cplx T[4]) {
    const cplx g = L.gam, g1 = L.gam1, g2 = L.gam2, g3 = L.gam3;
    const cplx ca = H.ca, cb = H.cb, xa = H.xa, ya = H.ya, xb = H.xb, yb = H.yb;
    const cplx kh = L.h * k;
    const cplx dc = ca - cb;
    // Every dA/dm has A's own pattern (a33 = a22, a34 = -a12, a42 = -a31, a43 = -a21, a44 = a11, a24 = -a13), so the
    // outer product r_a y_b is folded ONCE into ten combinations shared by the four parameter classes; the entries
    // of the third / fourth column and row carry 1/(2 mu) / 2 mu, applied to the combinations instead.
    const cplx c11 = r.v[0] * y.v[0] + r.v[3] * y.v[3];
    const cplx c22 = r.v[1] * y.v[1] + r.v[2] * y.v[2];
    const cplx c12 = r.v[0] * y.v[1] - r.v[2] * y.v[3];
    const cplx c21 = r.v[1] * y.v[0] - r.v[3] * y.v[2];
    const cplx i13 = L.imu2 * (r.v[0] * y.v[2] - r.v[1] * y.v[3]);
    const cplx i14 = L.imu2 * (r.v[0] * y.v[3]);
    const cplx i23 = L.imu2 * (r.v[1] * y.v[2]);
    const cplx m31 = L.mu2 * (r.v[2] * y.v[0] - r.v[3] * y.v[1]);
    const cplx m32 = L.mu2 * (r.v[2] * y.v[1]);
    const cplx m41 = L.mu2 * (r.v[3] * y.v[0]);
    const cplx g1sq = g1 * g1;
    // ---- rho (ipars = 1, :847-855): entries gamma/(2 rho mu) (.) and 2 mu gamma / rho (.) ----
    if (RHO) {
        cplx t = dc * (i13 + g1 * m31) + (ya - xb) * i14 + (yb - xa) * i23 + (g1sq * ya - xb) * m32 + (g1sq * yb - xa) * m41;
        T[0] = (g / L.rho) * t;
    }
    // ---- vp (ipars = 2, :829-844) ----
    {
        cplx ga = g2 * L.ia;
        cplx khca = kh * ca;
        cplx P1 = (kh * ya) * ga;
        cplx P2 = ((khca - ya) * ga) * L.iva2;
        cplx P3 = (khca + ya) * ga;
        cplx A1 = c11 - i13 + g1 * (m31 - c22);
        cplx A2 = g1 * c12 - i14 + g1sq * m32;
        cplx A3 = i23 - c21 - m41;
        T[1] = L.sca * (P1 * A1 + P2 * A2 + P3 * A3);
    }
    // ---- vs (ipars = 3, :811-826): common factor 2/beta ----
    {
        cplx khyb = kh * yb, khcb = kh * cb;
        cplx e1 = khcb + yb, e2 = khcb - yb;
        cplx gdc = g * dc, g1khyb = g1 * khyb, g1g3 = g1 * g3;
        cplx x11 = gdc - g1khyb;
        cplx x12 = g * (ya - xb) - e1;
        cplx x22 = khyb - gdc;
        cplx x21 = (yb - xa) + g1g3 * e2;
        cplx x31 = (2.0 * g - 1.0) * dc - g1khyb;
        cplx x32 = (2.0 * g) * (g1 * ya - xb) - e1;
        cplx x41 = 2.0 * (g1 * yb) - 2.0 * xa + (g1 * g1g3) * e2;
        cplx t = x11 * c11 + x12 * c12 + x22 * c22 + khyb * i13 + e1 * i14 - (e2 * (g * g3)) * i23 + x31 * m31 + x32 * m32
                 + g * (x21 * c21 + x41 * m41);
        T[2] = (L.scb * (2.0 * L.ib)) * t;
    }
    // ---- thickness (ipars = 4, :858-874) ----
    {
        cplx na = (H.sa * k) * (L.va * L.va);         // v_alpha * va_k
        cplx nb = (H.sb * k) * (L.vb * L.vb);
        cplx kca = k * ca, kcb = k * cb, nbcb = nb * cb, naca = na * ca;
        cplx kdx = (xa - xb) * k;
        cplx t = ((xa - g1 * xb) * k) * c11 + (g1 * kca - nbcb) * c12 - kdx * i13 + (nbcb - kca) * i14
                 + (g1 * kcb - naca) * c21 + ((xb - g1 * xa) * k) * c22 + (naca - kcb) * i23
                 + (g1 * kdx) * m31 + (g1sq * kca - nbcb) * m32 + (g1sq * kcb - naca) * m41;
        T[3] = g * t;
    }
}

// The density enters a layer matrix only through mu = rho beta^2, and mu only as the similarity A = D^-1 A' D with
// D = diag(mu, mu, 1, 1) (entries (1:2, 3:4) carry 1/mu, entries (3:4, 1:2) carry mu), so dA/drho = (A K - K A) / rho with
// K = diag(1, 1, 0, 0), and   r . dA/drho . y = ((r A)_{1:2} . y_{1:2} - r_{1:2} . (A y)_{1:2}) / rho:
// with the row above (ra = r A) and the column below (ya = A y) at hand, four products.  Only the real part is used.
RFS_HD double rf_rho_partial(const RfLayer& L, const V4& ra, const V4& y, const V4& r, const V4& ya) {
    return ((re_mul(ra.v[0], y.v[0]) + re_mul(ra.v[1], y.v[1])) - (re_mul(r.v[0], ya.v[0]) + re_mul(r.v[1], ya.v[1]))) / L.rho;
}

// e_row . (dE^-1/dm) . y for the half-space (RFModule.f90:924-987).  For rf_type 1
// the vp partial is identically zero, as in the reference; for rf_type 2 the
// reference multiplies by an unassigned variable (:933,980) -- here the evidently
// intended 1/va_k^3 is used (documented in DESIGN.md).
RFS_HD void rf_half_partials(const RfLayer& L, cplx omega, int rf_type, const V4& y, cplx T[4]) {
    cplx ta = omega * L.pva, tb = omega * L.pvb;
    double sa = (ta.re > 0.0 || (ta.re == 0.0 && ta.im >= 0.0)) ? 1.0 : -1.0;
    double sb = (tb.re > 0.0 || (tb.re == 0.0 && tb.im >= 0.0)) ? 1.0 : -1.0;
    cplx koa = sa * L.iva, kob = sb * L.ivb;          // k / v_alpha, k / v_beta
    if (omega.re == 0.0 && omega.im == 0.0) {
        // DC of the time-domain method: the reference forms gamma = 2 k^2 beta^2 / omega^2 = 0/0 (RFModule.f90:941),
        // every half-space partial becomes NaN and is zeroed (:698-703); the callers' NaN scrub does the same
        T[0] = T[1] = T[2] = C(__builtin_nan(""), __builtin_nan("")); T[3] = C(0.0);
        return;
    }
    cplx g = L.gam, g1 = L.gam1, g3 = L.gam3;
    cplx frho = (g * (0.5 * L.imu2)) / L.rho;         // gamma / (4 rho mu)
    cplx fb = g * L.ib;
    if (rf_type == 1) {
        T[0] = frho * (kob * y.v[2] + y.v[3]);
        T[1] = C(0.0);
        T[2] = L.scb * (fb * ((kob * (1.0 - g1 * g3)) * y.v[0] + y.v[1] + ((kob * g3) * L.imu2) * y.v[2]));
    } else {
        T[0] = frho * (-y.v[2] - koa * y.v[3]);
        cplx fa = (inv(L.ib * L.ib) * (L.ia * L.ia * L.ia)) * (L.iva2 * L.iva);
        T[1] = L.sca * (fa * (g1 * y.v[1] - L.imu2 * y.v[3]));
        T[2] = L.scb * (fb * (-y.v[0] - koa * y.v[1]));
    }
    T[3] = C(0.0);
}

// float32 pi of the reference's frequency axis: `pi = atan(1.0)*4.0` (RFModule.f90:364)
constexpr double RF_PI32 = 3.1415927410125732;

RFS_HD int rf_nextpow2(int n) { int m = 1; while (m < n) m *= 2; return m; }

}  // namespace rfs
