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

struct RfLayer {          // frequency-independent constants of one finite layer / the half-space
    cplx pva, pvb;        // sqrt(p^2 - 1/alpha^2), sqrt(p^2 - 1/beta^2)  (= p*va_k, p*vb_k)
    cplx va, vb, iva, ivb;// va_k, vb_k and reciprocals
    cplx iva2;            // 1/va_k^2
    cplx mu2, imu2;       // 2*mu, 1/(2*mu)
    cplx gam, gam1, gam2, gam3;
    cplx ia, ib;          // 1/alpha, 1/beta
    cplx sca, scb;        // alpha/vp, beta/vs  (RFModule.f90:624-628)
    double h, rho;
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
}

struct RfHyp {            // per (layer, frequency): cosh and the four scaled sinh terms
    cplx ca, cb, xa, ya, xb, yb;
    double sa, sb;        // branch sign: v_alpha = sa * omega * pva (principal sqrt, :729)
};

// nu*h = s*omega*pva*h with s chosen so that nu is the principal square root.
RFS_HD void rf_hyp(const RfLayer& L, cplx omega, RfHyp& H) {
    cplx ta = omega * L.pva, tb = omega * L.pvb;
    H.sa = (ta.re > 0.0 || (ta.re == 0.0 && ta.im >= 0.0)) ? 1.0 : -1.0;
    H.sb = (tb.re > 0.0 || (tb.re == 0.0 && tb.im >= 0.0)) ? 1.0 : -1.0;
    // cosh / sinh of the complex arguments from ONE exp, one reciprocal and one sincos each:
    // e^{a+ib} = e^a (c + i s), e^{-(a+ib)} = e^{-a} (c - i s)  ->  cosh = (ch c, sh s), sinh = (sh c, ch s)
    double e1 = fm_exp(L.h * ta.re), s1, c1, e2 = fm_exp(L.h * tb.re), s2, c2;
    fm_sincos(L.h * ta.im, &s1, &c1);
    fm_sincos(L.h * tb.im, &s2, &c2);
    double i1 = rcp_p(e1), i2 = rcp_p(e2);
    double ch1 = 0.5 * (e1 + i1), sh1 = 0.5 * (e1 - i1), ch2 = 0.5 * (e2 + i2), sh2 = 0.5 * (e2 - i2);
    H.ca = cplx{ch1 * c1, sh1 * s1}; H.cb = cplx{ch2 * c2, sh2 * s2};
    cplx sha = H.sa * cplx{sh1 * c1, ch1 * s1}, shb = H.sb * cplx{sh2 * c2, ch2 * s2};
    H.xa = L.va * sha; H.ya = sha * L.iva;
    H.xb = L.vb * shb; H.yb = shb * L.ivb;
}

struct V4 { cplx v[4]; };

// Haskell layer matrix entries (RFModule.f90:746-763), 10 distinct values.
struct RfA { cplx a11, a12, a13, a14, a21, a22, a23, a24, a31, a32, a41; };

RFS_HD void rf_build_A(const RfLayer& L, const RfHyp& H, RfA& A) {
    cplx g = L.gam, g1 = L.gam1;
    cplx dc = H.ca - H.cb;
    A.a11 = g * (H.ca - g1 * H.cb);
    A.a12 = g * (g1 * H.ya - H.xb);
    A.a13 = g * (-(dc * L.imu2));
    A.a14 = g * ((H.xb - H.ya) * L.imu2);
    A.a21 = g * (g1 * H.yb - H.xa);
    A.a22 = g * (H.cb - g1 * H.ca);
    A.a23 = g * ((H.xa - H.yb) * L.imu2);
    A.a24 = g * (dc * L.imu2);
    A.a31 = g * (L.mu2 * (g1 * dc));
    A.a32 = g * (L.mu2 * (g1 * g1 * H.ya - H.xb));
    A.a41 = g * (L.mu2 * (g1 * g1 * H.yb - H.xa));
    // a33 = a22, a34 = -a12, a42 = -a31, a43 = -a21, a44 = a11
}

RFS_HD V4 rf_row_times_A(const V4& r, const RfA& A) {   // r' = r . A
    V4 o;
    o.v[0] = r.v[0] * A.a11 + r.v[1] * A.a21 + r.v[2] * A.a31 + r.v[3] * A.a41;
    o.v[1] = r.v[0] * A.a12 + r.v[1] * A.a22 + r.v[2] * A.a32 - r.v[3] * A.a31;
    o.v[2] = r.v[0] * A.a13 + r.v[1] * A.a23 + r.v[2] * A.a22 - r.v[3] * A.a21;
    o.v[3] = r.v[0] * A.a14 + r.v[1] * A.a24 - r.v[2] * A.a12 + r.v[3] * A.a11;
    return o;
}

RFS_HD V4 rf_A_times_col(const RfA& A, const V4& y) {   // y' = A . y
    V4 o;
    o.v[0] = A.a11 * y.v[0] + A.a12 * y.v[1] + A.a13 * y.v[2] + A.a14 * y.v[3];
    o.v[1] = A.a21 * y.v[0] + A.a22 * y.v[1] + A.a23 * y.v[2] + A.a24 * y.v[3];
    o.v[2] = A.a31 * y.v[0] + A.a32 * y.v[1] + A.a22 * y.v[2] - A.a12 * y.v[3];
    o.v[3] = A.a41 * y.v[0] - A.a31 * y.v[1] - A.a21 * y.v[2] + A.a11 * y.v[3];
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
RFS_HD void rf_layer_partials(const RfLayer& L, const RfHyp& H, cplx k, const V4& r,
                              const V4& y, cplx T[4]) {
    const cplx g = L.gam, g1 = L.gam1, g2 = L.gam2, g3 = L.gam3;
    const cplx mu2 = L.mu2, imu2 = L.imu2;
    const cplx ca = H.ca, cb = H.cb, xa = H.xa, ya = H.ya, xb = H.xb, yb = H.yb;
    const cplx kh = L.h * k;
    const cplx dc = ca - cb;
    cplx z0, z1, z2, z3;
    // ---- rho (ipars = 1, :847-855) ----
    {
        cplx f = g * imu2 / L.rho;          // gamma / (2 rho mu)
        cplx q = (mu2 * g) / L.rho;         // 2 mu gamma / rho
        cplx g1sq = g1 * g1;
        cplx d13 = f * dc, d14 = f * (ya - xb), d23 = f * (yb - xa), d24 = -(f * dc);
        cplx d31 = q * (g1 * dc), d32 = q * (g1sq * ya - xb), d41 = q * (g1sq * yb - xa), d42 = -d31;
        z0 = d13 * y.v[2] + d14 * y.v[3];
        z1 = d23 * y.v[2] + d24 * y.v[3];
        z2 = d31 * y.v[0] + d32 * y.v[1];
        z3 = d41 * y.v[0] + d42 * y.v[1];
        T[0] = r.v[0] * z0 + r.v[1] * z1 + r.v[2] * z2 + r.v[3] * z3;
    }
    // ---- vp (ipars = 2, :829-844) ----
    {
        cplx ga = g2 * L.ia;
        cplx P1 = (kh * ya) * ga;
        cplx P2 = ((kh * ca - ya) * ga) * L.iva2;
        cplx P3 = (kh * ca + ya) * ga;
        cplx g1P1 = g1 * P1, g1P2 = g1 * P2;
        z0 = P1 * y.v[0] + g1P2 * y.v[1] - (P1 * imu2) * y.v[2] - (P2 * imu2) * y.v[3];
        z1 = -(P3 * y.v[0]) - g1P1 * y.v[1] + (P3 * imu2) * y.v[2] + (P1 * imu2) * y.v[3];
        z2 = (mu2 * g1P1) * y.v[0] + (mu2 * (g1 * g1P2)) * y.v[1] - g1P1 * y.v[2] - g1P2 * y.v[3];
        z3 = -((mu2 * P3) * y.v[0]) - (mu2 * g1P1) * y.v[1] + P3 * y.v[2] + P1 * y.v[3];
        T[1] = L.sca * (r.v[0] * z0 + r.v[1] * z1 + r.v[2] * z2 + r.v[3] * z3);
    }
    // ---- vs (ipars = 3, :811-826) ----
    {
        cplx b = L.ib, b2 = 2.0 * b;
        cplx imub = (2.0 * imu2) * b;                 // 1/(mu beta)
        cplx khyb = kh * yb, khcb = kh * cb;
        cplx e1 = khcb + yb, e2 = khcb - yb;
        cplx d11 = b2 * (g * dc - g1 * khyb);
        cplx d12 = b2 * (g * (ya - xb) - e1);
        cplx d13 = khyb * imub;
        cplx d14 = e1 * imub;
        cplx d21 = ((yb - xa) + (g1 * g3) * e2) * (g * b2);
        cplx d22 = b2 * (khyb - g * dc);
        cplx d23 = -((e2 * (g * g3)) * imub);
        cplx mb4 = (2.0 * mu2) * b;                   // 4 mu / beta
        cplx d31 = mb4 * ((2.0 * g - 1.0) * dc - g1 * khyb);
        cplx d32 = mb4 * ((2.0 * g) * (g1 * ya - xb) - e1);
        cplx d41 = (mb4 * g) * (2.0 * (g1 * yb) - 2.0 * xa + (g1 * g1 * g3) * e2);
        z0 = d11 * y.v[0] + d12 * y.v[1] + d13 * y.v[2] + d14 * y.v[3];
        z1 = d21 * y.v[0] + d22 * y.v[1] + d23 * y.v[2] - d13 * y.v[3];
        z2 = d31 * y.v[0] + d32 * y.v[1] + d22 * y.v[2] - d12 * y.v[3];
        z3 = d41 * y.v[0] - d31 * y.v[1] - d21 * y.v[2] + d11 * y.v[3];
        T[2] = L.scb * (r.v[0] * z0 + r.v[1] * z1 + r.v[2] * z2 + r.v[3] * z3);
    }
    // ---- thickness (ipars = 4, :858-874) ----
    {
        cplx na = (H.sa * k) * (L.va * L.va);         // v_alpha * va_k
        cplx nb = (H.sb * k) * (L.vb * L.vb);
        cplx kca = k * ca, kcb = k * cb, nbcb = nb * cb, naca = na * ca;
        cplx h11 = (xa - g1 * xb) * k;
        cplx h12 = g1 * kca - nbcb;
        cplx h13 = ((xb - xa) * k) * imu2;
        cplx h14 = (nbcb - kca) * imu2;
        cplx h21 = g1 * kcb - naca;
        cplx h22 = (xb - g1 * xa) * k;
        cplx h23 = (naca - kcb) * imu2;
        cplx h31 = (mu2 * g1) * (k * (xa - xb));
        cplx h32 = mu2 * (g1 * g1 * kca - nbcb);
        cplx h41 = mu2 * (g1 * g1 * kcb - naca);
        z0 = h11 * y.v[0] + h12 * y.v[1] + h13 * y.v[2] + h14 * y.v[3];
        z1 = h21 * y.v[0] + h22 * y.v[1] + h23 * y.v[2] - h13 * y.v[3];
        z2 = h31 * y.v[0] + h32 * y.v[1] + h22 * y.v[2] - h12 * y.v[3];
        z3 = h41 * y.v[0] - h31 * y.v[1] - h21 * y.v[2] + h11 * y.v[3];
        T[3] = g * (r.v[0] * z0 + r.v[1] * z1 + r.v[2] * z2 + r.v[3] * z3);
    }
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
