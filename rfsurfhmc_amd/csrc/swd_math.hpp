// Rayleigh-wave dispersion math for one lane: Dunkin secular function, the
// reference-semantics root search as a request/advance state machine, and the
// eigenfunction / Frechet-kernel sweeps.
//
// What it reproduces (reference file:line under src/SWD):
//   secular function   surfdisp96.f:791-1088 (dltar4, var, dnka, normc)
//   root search        surfdisp96.f:54-368, 375-396, 398-491, 568-701 and the
//                      per-period retry of surfdisp.cpp:93-100
//   eigenfunctions     sregn96.f90:196-402 (svfunc), 404-492 (up), 993-1063 (down),
//                      494-650 (dnka), 917-991 (hska), 831-915 (varsv), 652-829 (evalg)
//   energy / kernels   sregn96.f90:1065-1201 (energy), 1203-1323 (intijr),
//                      1325-1403 (f/g/h1/h2), 1436-1535 (getdcdh), 1537-1589 (getmat)
//
// How it is re-designed for a 64-wide wavefront:
//   * the root search is a state machine (RootSearch::advance) that only ever
//     REQUESTS "evaluate the secular function at c"; the kernel evaluates it for all
//     64 lanes at one call site, so lanes in different phases (bracketing scan,
//     bisection, Neville step, different periods, retry mode) never diverge inside
//     the expensive layer loop;
//   * the Neville table lives in registers (predicated static-index updates);
//   * the eigenfunction pass stores only the compound up-sweep (6 doubles / layer) in a
//     coalesced HBM scratch and fuses down-sweep + eigenfunctions + energy integrals +
//     interface terms into ONE top-down sweep with O(1) state; the six layer integrals
//     share one E/E^-1/potential evaluation (the reference re-runs evalg six times).
//
// Model values are float32 in memory (the reference force-casts its inputs,
// src/SWD/main.cpp:9) and are widened to f64 on load, like dble(b(m)) in the Fortran.
#pragma once
#include "cplx.hpp"

namespace rfs {

// strided float32 layer arrays: element m of lane's model at ptr[m * stride]
struct SwdModel {
    const float* d; const float* a; const float* b; const float* rho;
    long stride;
    int n;
    RFS_HD double D(int m) const { return (double)d[m * stride]; }
    RFS_HD double A(int m) const { return (double)a[m * stride]; }
    RFS_HD double B(int m) const { return (double)b[m * stride]; }
    RFS_HD double R(int m) const { return (double)rho[m * stride]; }
    RFS_HD float Bf(int m) const { return b[m * stride]; }
    RFS_HD float Af(int m) const { return a[m * stride]; }
    RFS_HD float Rf(int m) const { return rho[m * stride]; }
};

// f64 layer arrays of an earth-flattened model (bldsph, sregn96.f90:133-187 / slegn96.f90:107-167);
// the float32 views are sngl(zrho(m)), sngl(zb(m)) as the Fortran passes them to dnka / hska.
struct SwdModelD {
    const double* d; const double* a; const double* b; const double* rho;
    long stride;
    int n;
    RFS_HD double D(int m) const { return d[m * stride]; }
    RFS_HD double A(int m) const { return a[m * stride]; }
    RFS_HD double B(int m) const { return b[m * stride]; }
    RFS_HD double R(int m) const { return rho[m * stride]; }
    RFS_HD float Bf(int m) const { return (float)b[m * stride]; }
    RFS_HD float Rf(int m) const { return (float)rho[m * stride]; }
};

RFS_HD double sgn1(double x) { return copysign(1.0, x); }
// sgn1(a) != sgn1(b), i.e. sgn1(a) * sgn1(b) < 0: the sign bits differ (NaN and -0 by their sign bit, like copysign)
RFS_HD bool diffsign(double a, double b) { return (bool)signbit(a) != (bool)signbit(b); }

// ---------------------------------------------------------------------------
// Love secular function (surfdisp96.f:727-787 dltar1, all-solid model): 2-vector recurrence
// from the half-space up, normalised per layer; e(1) at the surface is the period equation.
// ---------------------------------------------------------------------------
RFS_HD double swd_secular_love(const SwdModel& M, double wvno, double omega) {
    const int last = M.n - 1;
    double beta1 = M.B(last), rho1 = M.R(last);
    double xkb = omega / beta1;
    double rb = sqrt((wvno + xkb) * fabs(wvno - xkb));
    double e1 = rho1 * rb, e2 = 1.0 / (beta1 * beta1);
    const int llw = (M.B(0) <= 0.0) ? 1 : 0;        // water on top (surfdisp96.f:138-139): SH waves stop at its base (:750)
    for (int m = last - 1; m >= llw; m--) {
        beta1 = M.B(m); rho1 = M.R(m);
        double xmu = rho1 * beta1 * beta1;
        xkb = omega / beta1;
        rb = sqrt((wvno + xkb) * fabs(wvno - xkb));
        double dm = M.D(m), q = dm * rb, y, z, cosq;
        if (wvno < xkb) {
            double sn, cs; sincos(q, &sn, &cs);
            y = sn / rb; z = -rb * sn; cosq = cs;
        } else if (wvno == xkb) {
            cosq = 1.0; y = dm; z = 0.0;
        } else {
            double fac = (q < 16.0) ? exp(-2.0 * q) : 0.0;
            cosq = (1.0 + fac) * 0.5;
            double sinq = (1.0 - fac) * 0.5;
            y = sinq / rb; z = rb * sinq;
        }
        double e10 = e1 * cosq + e2 * xmu * z;
        double e20 = e1 * y / xmu + e2 * cosq;
        double xnor = fmax(fabs(e10), fabs(e20));
        if (xnor < 1.0e-40) xnor = 1.0;
        e1 = e10 / xnor; e2 = e20 / xnor;
    }
    return e1;
}

// ---------------------------------------------------------------------------
// Dunkin secular function  (surfdisp96.f:791-891 with var :894-1011, dnka :1044-1088,
// normc :1015-1040 fused per layer).  Only e(1) is returned; its sign drives the search.
// ---------------------------------------------------------------------------
RFS_HD double swd_secular(const SwdModel& M, double wvno, double omga) {
    double omega = omga < 1.0e-4 ? 1.0e-4 : omga;
    const double wvno2 = wvno * wvno;
    const int last = M.n - 1;
    double e0, e1, e2, e3, e4;
    {
        double xka = omega / M.A(last), xkb = omega / M.B(last);
        double ra = sqrt((wvno + xka) * fabs(wvno - xka));
        double rb = sqrt((wvno + xkb) * fabs(wvno - xkb));
        double t = M.B(last) / omega;
        double gammk = 2.0 * t * t, gam = gammk * wvno2, gamm1 = gam - 1.0;
        double rho1 = M.R(last);
        e0 = rho1 * rho1 * (gamm1 * gamm1 - gam * gammk * ra * rb);
        e1 = -rho1 * ra;
        e2 = rho1 * (gamm1 - gammk * ra * rb);
        e3 = rho1 * rb;
        e4 = wvno2 - ra * rb;
    }
    const int llw = (M.B(0) <= 0.0) ? 1 : 0;        // water on top, surfdisp96.f:138-139
    for (int m = last - 1; m >= llw; m--) {
        double xka = omega / M.A(m), xkb = omega / M.B(m);
        double t = M.B(m) / omega;
        double gammk = 2.0 * t * t, gam = gammk * wvno2;
        double ra = sqrt((wvno + xka) * fabs(wvno - xka));
        double rb = sqrt((wvno + xkb) * fabs(wvno - xkb));
        double dpth = M.D(m), rho = M.R(m);
        double p = ra * dpth, q = rb * dpth;
        // ---- var ----
        double pex = 0.0, sex = 0.0, cosp, w, x, cosq, y, z;
        if (wvno < xka) {
            double s, c; sincos(p, &s, &c);
            w = s / ra; x = -ra * s; cosp = c;
        } else if (wvno == xka) {
            cosp = 1.0; w = dpth; x = 0.0;
        } else {
            pex = p;
            double fac = (p < 16.0) ? exp(-2.0 * p) : 0.0;
            cosp = (1.0 + fac) * 0.5;
            double sinp = (1.0 - fac) * 0.5;
            w = sinp / ra; x = ra * sinp;
        }
        if (wvno < xkb) {
            double s, c; sincos(q, &s, &c);
            y = s / rb; z = -rb * s; cosq = c;
        } else if (wvno == xkb) {
            cosq = 1.0; y = dpth; z = 0.0;
        } else {
            sex = q;
            double fac = (q < 16.0) ? exp(-2.0 * q) : 0.0;
            cosq = (1.0 + fac) * 0.5;
            double sinq = (1.0 - fac) * 0.5;
            y = sinq / rb; z = rb * sinq;
        }
        double exa = pex + sex;
        double a0 = (exa < 60.0) ? exp(-exa) : 0.0;
        double cpcq = cosp * cosq, cpy = cosp * y, cpz = cosp * z, cqw = cosq * w, cqx = cosq * x;
        double xy = x * y, xz = x * z, wy = w * y, wz = w * z;
        // ---- dnka ----
        double gamm1 = gam - 1.0, twgm1 = gam + gamm1, gmgmk = gam * gammk, gmgm1 = gam * gamm1;
        double gm1sq = gamm1 * gamm1, rho2 = rho * rho, a0pq = a0 - cpcq;
        double c11 = cpcq - 2.0 * gmgm1 * a0pq - gmgmk * xz - wvno2 * gm1sq * wy;
        double c12 = (wvno2 * cpy - cqx) / rho;
        double c13 = -(twgm1 * a0pq + gammk * xz + wvno2 * gamm1 * wy) / rho;
        double c14 = (cpz - wvno2 * cqw) / rho;
        double c15 = -(2.0 * wvno2 * a0pq + xz + wvno2 * wvno2 * wy) / rho2;
        double c21 = (gmgmk * cpz - gm1sq * cqw) * rho;
        double c22 = cpcq;
        double c23 = gammk * cpz - gamm1 * cqw;
        double c24 = -wz;
        double c41 = (gm1sq * cpy - gmgmk * cqx) * rho;
        double c42 = -xy;
        double c43 = gamm1 * cpy - gammk * cqx;
        double c51 = -(2.0 * gmgmk * gm1sq * a0pq + gmgmk * gmgmk * xz + gm1sq * gm1sq * wy) * rho2;
        double c53 = -(gammk * gamm1 * twgm1 * a0pq + gam * gammk * gammk * xz + gamm1 * gm1sq * wy) * rho;
        double tt = -2.0 * wvno2;
        double c31 = tt * c53, c32 = tt * c43, c33 = a0 + 2.0 * (cpcq - c11), c34 = tt * c23, c35 = tt * c13;
        // ee(i) = sum_j e(j) ca(j,i); ca(2,5)=ca(1,4), ca(4,4)=ca(2,2), ca(4,5)=ca(1,2),
        // ca(5,2)=ca(4,1), ca(5,4)=ca(2,1), ca(5,5)=ca(1,1)
        double n0 = e0 * c11 + e1 * c21 + e2 * c31 + e3 * c41 + e4 * c51;
        double n1 = e0 * c12 + e1 * c22 + e2 * c32 + e3 * c42 + e4 * c41;
        double n2 = e0 * c13 + e1 * c23 + e2 * c33 + e3 * c43 + e4 * c53;
        double n3 = e0 * c14 + e1 * c24 + e2 * c34 + e3 * c22 + e4 * c21;
        double n4 = e0 * c15 + e1 * c14 + e2 * c35 + e3 * c12 + e4 * c11;
        // ---- normc ----
        double t1 = fmax(fmax(fmax(fabs(n0), fabs(n1)), fmax(fabs(n2), fabs(n3))), fabs(n4));
        if (t1 < 1.0e-40) t1 = 1.0;
        e0 = n0 / t1; e1 = n1 / t1; e2 = n2 / t1; e3 = n3 / t1; e4 = n4 / t1;
    }
    if (llw) {                                       // the water layer, surfdisp96.f:870-886 (only the P part of var)
        double xka = omega / M.A(0);
        double ra = sqrt((wvno + xka) * fabs(wvno - xka));
        double dpth = M.D(0), p = ra * dpth, cosp, w;
        if (wvno < xka) {
            double s, c; sincos(p, &s, &c);
            w = s / ra; cosp = c;
        } else if (wvno == xka) {
            cosp = 1.0; w = dpth;
        } else {
            double fac = (p < 16.0) ? exp(-2.0 * p) : 0.0;
            cosp = (1.0 + fac) * 0.5;
            w = ((1.0 - fac) * 0.5) / ra;
        }
        return cosp * e0 + (-M.R(0) * w) * e1;
    }
    return e0;
}

// ---------------------------------------------------------------------------
// Split form of the same secular function for "several lanes per chain" root searches:
//   swd_layer_entries  -- everything of one layer that does NOT depend on the propagated
//                         vector (var + dnka): 15 numbers, computed by any lane;
//   swd_halfspace_e    -- the half-space start vector;
//   swd_apply_layer    -- e <- normc(e . CA), the only sequential part (25 FMA + normalise).
// Divisions are replaced by per-layer reciprocals (SwdLayerC) and rsqrt; results agree with
// swd_secular to a few ulp (the root search only needs the sign and the smooth magnitude).
// ---------------------------------------------------------------------------
struct SwdLayerC { double d, ia, ib, b, rho, irho; };   // thickness, 1/alpha, 1/beta, beta, rho, 1/rho

constexpr int SWD_NENT = 15;

RFS_HD void swd_trig_split(double wvno, double xk, double dpth, double& ex, double& cosx, double& w, double& x,
                           double& eh) {
    // var (surfdisp96.f:941-1002) for one wave type: returns cos, sin/r, +-r*sin, the exponent ex and
    // eh = exp(-ex) (1 when oscillatory): exp(-2p) is formed as eh*eh and the layer's exp(-(pex+sex)) as the
    // product of the two eh, so an evanescent wave type costs ONE exponential in all.
    const double v = (wvno + xk) * fabs(wvno - xk);
    const bool osc = wvno < xk;
    const double ir = rsqrt_p(v), r = v * ir, p = r * dpth;
    double cs, sn;
    eh = 1.0;
    if (osc) {
        fm_sincos(p, &sn, &cs);
    } else {
        eh = fm_exp(-p);
        double fac = (p < 16.0) ? eh * eh : 0.0;
        cs = (1.0 + fac) * 0.5;
        sn = (1.0 - fac) * 0.5;
    }
    cosx = cs;
    w = sn * ir;
    x = (osc ? -r : r) * sn;
    ex = osc ? 0.0 : p;
    // wvno == xk exactly (v = 0: ir = inf, r = p = NaN above): the reference's middle branch.  On the device the
    // fix-up is skipped wave-wide unless some lane needs it, which keeps ~20 selects out of the producers' hot loop.
    const bool deg = (wvno == xk);
#if defined(__HIP_DEVICE_COMPILE__)
    if (__any(deg))
#endif
    {
        if (deg) { cosx = 1.0; w = dpth; x = 0.0; ex = 0.0; eh = 1.0; }
    }
}

// swd_trig_split for the P and the S wavenumber of one layer AT ONCE: the same numbers lane by lane, but the two square roots, the
// two exponentials and the S sine / cosine -- long chains of dependent f64 operations each -- stand side by side in ONE branch-free
// block, so that a wavefront that has its SIMD nearly to itself (k_swd_exact: 1.25 per SIMD) overlaps them.  What a lane does not
// need (the exponential of an oscillatory wave type, the sine of an evanescent one) is computed and dropped: more instructions, a
// shorter chain -- for kernels short of wavefronts only.  (The P sine / cosine stays behind a wavefront-wide test: phase
// velocities below the P velocities hardly ever need it.)
RFS_HD void swd_trig_split2(double wvno, double xka, double xkb, double dpth,
                            double& pex, double& cosp, double& w, double& x, double& eha,
                            double& sex, double& cosq, double& y, double& z, double& ehb, const FmVC* vc = nullptr) {
    const double va = (wvno + xka) * fabs(wvno - xka), vb = (wvno + xkb) * fabs(wvno - xkb);
    const bool osca = wvno < xka, oscb = wvno < xkb;
    const double ira = rsqrt_p(va), irb = rsqrt_p(vb);
    const double ra = va * ira, rb = vb * irb, pa = ra * dpth, pb = rb * dpth;
    double ea, eb;
    ea = fm_exp(-pa); eb = fm_exp(-pb);
    const double faca = (pa < 16.0) ? ea * ea : 0.0, facb = (pb < 16.0) ? eb * eb : 0.0;
    double csa = (1.0 + faca) * 0.5, sna = (1.0 - faca) * 0.5, csb = (1.0 + facb) * 0.5, snb = (1.0 - facb) * 0.5;
#if defined(__HIP_DEVICE_COMPILE__)
    if (__any(osca))
#endif
    { double sn, cs; if (vc) fm_sincos_vc(pa, *vc, &sn, &cs); else fm_sincos(pa, &sn, &cs); csa = osca ? cs : csa; sna = osca ? sn : sna; }
    { double sn, cs; if (vc) fm_sincos_vc(pb, *vc, &sn, &cs); else fm_sincos(pb, &sn, &cs); csb = oscb ? cs : csb; snb = oscb ? sn : snb; }
    eha = osca ? 1.0 : ea; ehb = oscb ? 1.0 : eb;
    cosp = csa; w = sna * ira; x = (osca ? -ra : ra) * sna; pex = osca ? 0.0 : pa;
    cosq = csb; y = snb * irb; z = (oscb ? -rb : rb) * snb; sex = oscb ? 0.0 : pb;
    const bool dega = (wvno == xka), degb = (wvno == xkb);
#if defined(__HIP_DEVICE_COMPILE__)
    if (__any(dega || degb))
#endif
    {
        if (dega) { cosp = 1.0; w = dpth; x = 0.0; pex = 0.0; eha = 1.0; }
        if (degb) { cosq = 1.0; y = dpth; z = 0.0; sex = 0.0; ehb = 1.0; }
    }
}

template <bool DUAL = false>
RFS_HD void swd_layer_entries(const SwdLayerC& L, double wvno, double wvno2, double omega, double iomega,
                              double ent[SWD_NENT], const FmVC* vc = nullptr) {
    RFS_NO_CONTRACT
    double xka = omega * L.ia, xkb = omega * L.ib;
    double t = L.b * iomega;
    double gammk = 2.0 * t * t, gam = gammk * wvno2;
    double pex, sex, cosp, w, x, cosq, y, z, eha, ehb;
    if (DUAL) swd_trig_split2(wvno, xka, xkb, L.d, pex, cosp, w, x, eha, sex, cosq, y, z, ehb, vc);
    else {
        swd_trig_split(wvno, xka, L.d, pex, cosp, w, x, eha);
        swd_trig_split(wvno, xkb, L.d, sex, cosq, y, z, ehb);
    }
    double exa = pex + sex;
    double a0 = (exa < 60.0) ? eha * ehb : 0.0;
    double cpcq = cosp * cosq, cpy = cosp * y, cpz = cosp * z, cqw = cosq * w, cqx = cosq * x;
    double xy = x * y, xz = x * z, wy = w * y, wz = w * z;
    double gamm1 = gam - 1.0, twgm1 = gam + gamm1, gmgmk = gam * gammk, gmgm1 = gam * gamm1;
    double gm1sq = gamm1 * gamm1, rho = L.rho, ir = L.irho, rho2 = rho * rho, ir2 = ir * ir, a0pq = a0 - cpcq;
    // (every fused multiply-add written out, see RFS_NO_CONTRACT: left to right through the reference's expressions, dnka)
    double c11 = ::fma(-(2.0 * gmgm1), a0pq, cpcq);                                 // cpcq - 2 gmgm1 a0pq - gmgmk xz - wvno2 gm1sq wy
    c11 = ::fma(-gmgmk, xz, c11);
    c11 = ::fma(-(wvno2 * gm1sq), wy, c11);
    ent[0] = c11;
    ent[1] = ::fma(wvno2, cpy, -cqx) * ir;                                          // c12
    double s13 = twgm1 * a0pq;                                                      // c13 = -(twgm1 a0pq + gammk xz + wvno2 gamm1 wy) / rho
    s13 = ::fma(gammk, xz, s13);
    s13 = ::fma(wvno2 * gamm1, wy, s13);
    ent[2] = -s13 * ir;
    ent[3] = ::fma(-wvno2, cqw, cpz) * ir;                                          // c14
    double s15 = ::fma(2.0 * wvno2, a0pq, xz);                                      // c15 = -(2 wvno2 a0pq + xz + wvno2^2 wy) / rho^2
    s15 = ::fma(wvno2 * wvno2, wy, s15);
    ent[4] = -s15 * ir2;
    ent[5] = ::fma(gmgmk, cpz, -(gm1sq * cqw)) * rho;                               // c21
    ent[6] = cpcq;                                                                  // c22
    ent[7] = ::fma(gammk, cpz, -(gamm1 * cqw));                                     // c23
    ent[8] = -wz;                                                                   // c24
    ent[9] = ::fma(gm1sq, cpy, -(gmgmk * cqx)) * rho;                               // c41
    ent[10] = -xy;                                                                  // c42
    ent[11] = ::fma(gamm1, cpy, -(gammk * cqx));                                    // c43
    double s51 = ((2.0 * gmgmk) * gm1sq) * a0pq;                                    // c51 = -(2 gmgmk gm1sq a0pq + gmgmk^2 xz + gm1sq^2 wy) rho^2
    s51 = ::fma(gmgmk * gmgmk, xz, s51);
    s51 = ::fma(gm1sq * gm1sq, wy, s51);
    ent[12] = -s51 * rho2;
    double s53 = ((gammk * gamm1) * twgm1) * a0pq;                                  // c53 = -(gammk gamm1 twgm1 a0pq + gam gammk^2 xz + gamm1 gm1sq wy) rho
    s53 = ::fma((gam * gammk) * gammk, xz, s53);
    s53 = ::fma(gamm1 * gm1sq, wy, s53);
    ent[13] = -s53 * rho;
    ent[14] = ::fma(2.0, cpcq - c11, a0);                                           // c33
}

RFS_HD void swd_halfspace_e(const SwdLayerC& L, double wvno, double wvno2, double omega, double iomega, double e[5]) {
    RFS_NO_CONTRACT
    double xka = omega * L.ia, xkb = omega * L.ib;
    double ra = sqrt((wvno + xka) * fabs(wvno - xka));
    double rb = sqrt((wvno + xkb) * fabs(wvno - xkb));
    double t = L.b * iomega;
    double gammk = 2.0 * t * t, gam = gammk * wvno2, gamm1 = gam - 1.0, rho1 = L.rho;
    e[0] = (rho1 * rho1) * ::fma(-((gam * gammk) * ra), rb, gamm1 * gamm1);
    e[1] = -rho1 * ra;
    e[2] = rho1 * ::fma(-(gammk * ra), rb, gamm1);
    e[3] = rho1 * rb;
    e[4] = ::fma(-ra, rb, wvno2);
}

RFS_HD void swd_apply_layer(double e[5], const double c[SWD_NENT], double tt /* -2 wvno^2 */) {
    double e2t = e[2] * tt;
    double n0 = e[0] * c[0] + e[1] * c[5] + e2t * c[13] + e[3] * c[9] + e[4] * c[12];
    double n1 = e[0] * c[1] + e[1] * c[6] + e2t * c[11] + e[3] * c[10] + e[4] * c[9];
    double n2 = e[0] * c[2] + e[1] * c[7] + e[2] * c[14] + e[3] * c[11] + e[4] * c[13];
    double n3 = e[0] * c[3] + e[1] * c[8] + e2t * c[7] + e[3] * c[6] + e[4] * c[5];
    double n4 = e[0] * c[4] + e[1] * c[3] + e2t * c[2] + e[3] * c[1] + e[4] * c[0];
    double t1 = fmax(fmax(fmax(fabs(n0), fabs(n1)), fmax(fabs(n2), fabs(n3))), fabs(n4));
    if (t1 < 1.0e-40) t1 = 1.0;
    double it1 = rcp_p(t1);
    e[0] = n0 * it1; e[1] = n1 * it1; e[2] = n2 * it1; e[3] = n3 * it1; e[4] = n4 * it1;
}

// The per-layer normc (surfdisp96.f:1015-1040) only rescales the vector by a positive number, and the
// value the search uses is e(1) of the LAST normalised vector, i.e. v(1)/max|v| of the raw product
// v = e_half . CA_{n-2} ... CA_0: intermediate normalisations cancel exactly.  The raw recurrence
// therefore drops the max/reciprocal chain from every layer (25 FMAs remain); an exact power-of-two
// rescale every few layers keeps the range, and swd_finish applies the one normalisation that matters.
RFS_HD void swd_apply_layer_raw(double e[5], const double c[SWD_NENT], double tt /* -2 wvno^2 */) {
    RFS_NO_CONTRACT
    const double e2t = e[2] * tt;
    const double n0 = ::fma(e[4], c[12], ::fma(e[3], c[9], ::fma(e2t, c[13], ::fma(e[1], c[5], e[0] * c[0]))));
    const double n1 = ::fma(e[4], c[9], ::fma(e[3], c[10], ::fma(e2t, c[11], ::fma(e[1], c[6], e[0] * c[1]))));
    const double n2 = ::fma(e[4], c[13], ::fma(e[3], c[11], ::fma(e[2], c[14], ::fma(e[1], c[7], e[0] * c[2]))));
    const double n3 = ::fma(e[4], c[5], ::fma(e[3], c[6], ::fma(e2t, c[7], ::fma(e[1], c[8], e[0] * c[3]))));
    const double n4 = ::fma(e[4], c[0], ::fma(e[3], c[1], ::fma(e2t, c[2], ::fma(e[1], c[3], e[0] * c[4]))));
    e[0] = n0; e[1] = n1; e[2] = n2; e[3] = n3; e[4] = n4;
}
RFS_HD void swd_rescale_pow2(double e[5]) {
    double t1 = fmax(fmax(fmax(fabs(e[0]), fabs(e[1])), fmax(fabs(e[2]), fabs(e[3]))), fabs(e[4]));
    int ex = 0;
    if (t1 > 0.0 && t1 < 1.0e300) frexp(t1, &ex);
    e[0] = ldexp(e[0], -ex); e[1] = ldexp(e[1], -ex); e[2] = ldexp(e[2], -ex);
    e[3] = ldexp(e[3], -ex); e[4] = ldexp(e[4], -ex);
}
RFS_HD double swd_finish(const double e[5]) {
    double t1 = fmax(fmax(fmax(fabs(e[0]), fabs(e[1])), fmax(fabs(e[2]), fabs(e[3]))), fabs(e[4]));
    if (t1 < 1.0e-40) t1 = 1.0;
    return e[0] / t1;
}

// The two secular functions behind one interface for the lanes-per-item search (k_swd_roots_split): NENT numbers per
// layer that do not depend on the propagated vector, an NV-vector carried through the layers, the start vector of the
// half-space.  Rayleigh: the Dunkin form above.  Love (dltar1, surfdisp96.f:727-787): e <- e . [[cosq, y/mu], [mu z, cosq]]
// with the same var-style scaling of the evanescent branch; as for Rayleigh the per-layer normalisation only rescales by a
// positive number, so the raw recurrence with exact power-of-two rescales and one final e1 / max|e| gives the same
// Delta up to rounding.
struct SwdRayFamily {
    static constexpr int NENT = SWD_NENT, NV = 5;
    static constexpr bool LOVE = false;
    static RFS_HD void entries(const SwdLayerC& L, double wvno, double wvno2, double omega, double iomega, double* ent) {
        swd_layer_entries<false>(L, wvno, wvno2, omega, iomega, ent);
    }
    // the same numbers with the two wave types' functions side by side (swd_trig_split2): for kernels short of wavefronts
    static RFS_HD void entries_dual(const SwdLayerC& L, double wvno, double wvno2, double omega, double iomega, double* ent,
                                    const FmVC* vc = nullptr) {
        swd_layer_entries<true>(L, wvno, wvno2, omega, iomega, ent, vc);
    }
    static RFS_HD void halfspace(const SwdLayerC& L, double wvno, double wvno2, double omega, double iomega, double* e) {
        swd_halfspace_e(L, wvno, wvno2, omega, iomega, e);
    }
    static RFS_HD void apply(double* e, const double* c, double tt) { swd_apply_layer_raw(e, c, tt); }
    static RFS_HD void rescale(double* e) { swd_rescale_pow2(e); }
    static RFS_HD double finish(const double* e) { return swd_finish(e); }
};
struct SwdLoveFamily {
    static constexpr int NENT = 3, NV = 2;
    static constexpr bool LOVE = true;
    static RFS_HD void entries(const SwdLayerC& L, double wvno, double, double omega, double, double* ent) {
        double ex, cosq, y, z, eh;
        swd_trig_split(wvno, omega * L.ib, L.d, ex, cosq, y, z, eh);
        ent[0] = cosq;
        ent[1] = (L.rho * L.b * L.b) * z;                     // mu z
        ent[2] = y * (L.irho * L.ib * L.ib);                  // y / mu
    }
    static RFS_HD void entries_dual(const SwdLayerC& L, double wvno, double wvno2, double omega, double iomega, double* ent,
                                    const FmVC* = nullptr) {
        entries(L, wvno, wvno2, omega, iomega, ent);
    }
    static RFS_HD void halfspace(const SwdLayerC& L, double wvno, double, double omega, double, double* e) {
        const double xkb = omega * L.ib;
        e[0] = L.rho * sqrt((wvno + xkb) * fabs(wvno - xkb));
        e[1] = L.ib * L.ib;
    }
    static RFS_HD void apply(double* e, const double* c, double) {
        RFS_NO_CONTRACT
        const double n0 = ::fma(e[1], c[1], e[0] * c[0]), n1 = ::fma(e[1], c[0], e[0] * c[2]);
        e[0] = n0; e[1] = n1;
    }
    static RFS_HD void rescale(double* e) {
        const double t1 = fmax(fabs(e[0]), fabs(e[1]));
        int ex = 0;
        if (t1 > 0.0 && t1 < 1.0e300) frexp(t1, &ex);
        e[0] = ldexp(e[0], -ex); e[1] = ldexp(e[1], -ex);
    }
    static RFS_HD double finish(const double* e) {
        double t1 = fmax(fabs(e[0]), fabs(e[1]));
        if (t1 < 1.0e-40) t1 = 1.0;
        return e[0] / t1;
    }
};
template <int NV> RFS_HD void swd_rescale_pow2_n(double* e) {
    double t1 = fabs(e[0]);
#pragma unroll
    for (int j = 1; j < NV; j++) t1 = fmax(t1, fabs(e[j]));
    int ex = 0;
    if (t1 > 0.0 && t1 < 1.0e300) frexp(t1, &ex);
#pragma unroll
    for (int j = 0; j < NV; j++) e[j] = ldexp(e[j], -ex);
}
template <int NV> RFS_HD double swd_finish_n(const double* e) {
    double t1 = fabs(e[0]);
#pragma unroll
    for (int j = 1; j < NV; j++) t1 = fmax(t1, fabs(e[j]));
    if (t1 < 1.0e-40) t1 = 1.0;
    return e[0] / t1;
}

// surfdisp96.f:375-396  gtsolh (single precision throughout)
RFS_HD float swd_gtsolh(float a, float b) {
    float c = 0.95f * b;
    for (int i = 0; i < 5; i++) {
        float gamma = b / a, kappa = c / b;
        float k2 = kappa * kappa;
        float gk = gamma * kappa, gk2 = gk * gk;
        float fac1 = sqrtf(1.0f - gk2), fac2 = sqrtf(1.0f - k2);
        float fr = (2.0f - k2) * (2.0f - k2) - 4.0f * fac1 * fac2;
        float frp = -4.0f * (2.0f - k2) * kappa + 4.0f * fac2 * gamma * gamma * kappa / fac1 +
                    4.0f * fac1 * kappa / fac2;
        frp = frp / b;
        c = c - fr / frp;
    }
    return c;
}

// surfdisp96.f:149-160 extremal velocities, :203-222 start value of the search: 0.95 * 0.90 * (Rayleigh velocity of the
// half-space made of the slowest layer), all in single precision; betmx = fastest S velocity
constexpr double SWD_MAX_SCAN = 2000.0;      // km/s between the start value and the fastest layer: 4e5 cells of the scan (RootSearchT::begin)
template <class Mdl>
RFS_HD float swd_start_value(const Mdl& M, float& betmx) {
    float bmx = -1.e20f, bmn = 1.e20f; int jmn = 0, jsol = 1;
    for (int i = 0; i < M.n; i++) {
        float b = M.Bf(i), a = M.Af(i);
        if (b > 0.01f && b < bmn) { bmn = b; jmn = i; jsol = 1; }
        else if (b <= 0.01f && a < bmn) { bmn = a; jmn = i; jsol = 0; }
        if (b > bmx) bmx = b;
    }
    float cc1 = (jsol == 0) ? bmn : swd_gtsolh(M.Af(jmn), M.Bf(jmn));
    cc1 = 0.95f * cc1; cc1 = 0.90f * cc1;
    betmx = bmx;
    return cc1;
}

// ---------------------------------------------------------------------------
// Root search state machine.  Usage:
//     rs.begin(model, periods, kmax);
//     while (!rs.done) { double del = swd_secular(M, rs.omega / rs.creq, rs.omega); rs.advance(del); }
// Results: rs.flag (1 ok / 0 failed), cg[k] written through the Out functor as the
// float32-rounded phase velocity (surfdisp96.f:302,307), zeros after a failure.
// ---------------------------------------------------------------------------
// Neville table storage: registers (static-index predicated updates) by default; the cooperative
// kernel keeps it in LDS (dynamic indexing, 48 fewer live VGPRs across the divergent state machine).
struct NevTabReg {
    double x[12], y[12];
    static constexpr bool kDynamic = false;
    RFS_HD double gx(int i) const { return x[i]; }
    RFS_HD double gy(int i) const { return y[i]; }
    RFS_HD void sx(int i, double v) { x[i] = v; }
    RFS_HD void sy(int i, double v) { y[i] = v; }
};
struct NevTabMem {             // x(i) at base[i*stride], y(i) at base[(12+i)*stride]
    double* base; int stride;
    static constexpr bool kDynamic = true;
    RFS_HD double gx(int i) const { return base[i * stride]; }
    RFS_HD double gy(int i) const { return base[(12 + i) * stride]; }
    RFS_HD void sx(int i, double v) { base[i * stride] = v; }
    RFS_HD void sy(int i, double v) { base[(12 + i) * stride] = v; }
};

// omega of period k: a period functor may provide a precomputed table (omega(k)); otherwise 2 pi / T(k)
template <class F> RFS_HD auto rs_omega_of(const F& T, int k, int) -> decltype(T.omega(k)) { return T.omega(k); }
template <class F> RFS_HD double rs_omega_of(const F& T, int k, long) { return (2.0 * 3.141592653589793) / T(k); }

// MODES = true: the mode loop of surfdisp96.f:227-316 (`do 1800 iq=1,mode`) and its per-period retry
// (surfdisp.cpp:93-100) around the same getsol / nevill machine.  Mode iq searches above mode iq-1, whose unrounded
// roots c(k) it reads from -- and overwrites in -- a scratch (cmb); the float32 outputs are overwritten mode after mode,
// so the LAST mode's values remain; a mode not found from period k on zeroes the outputs from there and caps every
// later mode at that period (ift); only the fundamental's failure sets ierr and triggers the retry, which redoes every
// period whose output is zero / NaN on its own (all modes, each starting from the start value).  The Out functor must
// then also provide get(k).  MODES = false compiles the fundamental-only code unchanged.
template <class Tab = NevTabReg, bool MODES = false>
struct RootSearchT {
    enum { PH_START, PH_SCAN, PH_HALF0, PH_HALF_OUT, PH_HALF_B, PH_NEV };
    static constexpr double TWOPI = 2.0 * 3.141592653589793;
    // per-model constants
    double cc, dc, cm; float betmx;
    int kmax;
    // state
    int k, phase, ifirst, retry, idir, done, flag;
    double omega, creq;
    double c1, c2, del1, del2, clow, del1st, cprev;
    double c3, del3;
    int nev, m, nctrl;
    Tab tab;
    long nsec;
    // MODES only: mode index, number of modes, cap period of the later modes, fundamental failed; scratch of unrounded c(k)
    int iq = 0, nmode = 1, ift = 1 << 30, fund_failed = 0;
    int absurd = 0;            // the model's scan would be longer than SWD_MAX_SCAN (begin): the search fails at its first evaluation
    double* cmb = nullptr; long cms = 0;
    RFS_HD double kept(int kk) const { return cmb[(long)kk * cms]; }

    template <class PeriodFn>
    RFS_HD void start_period(const PeriodFn& T) {
        if (MODES && iq > 0) {
            const double one = 1.0e-2;                                 // `one` of surfdisp96.f:141
            if (retry || k == 0) { c1 = kept(k) + one * dc; clow = c1; ifirst = 1; }          // :261-264
            else { ifirst = 0; clow = kept(k) + one * dc; c1 = cprev; if (c1 < clow) c1 = clow; }   // :265-271
        }
        else if (retry || k == 0) { c1 = cc; clow = cc; ifirst = 1; }  // surfdisp96.f:257-260
        else { ifirst = 0; c1 = cprev - 1.5 * dc; clow = cm; }         // :272-275 (onea = 1.5)
        omega = rs_omega_of(T, k, 0);
        creq = c1; phase = PH_START;
    }

    // MODES: a period's search is over -- found (failed = false; k and iq not advanced yet) or not.  Sets up the next
    // search (next period, next mode, next retry period) or ends the call.  surfdisp96.f:236, 317-362 (labels 1700 /
    // 1750), surfdisp.cpp:93-100.
    template <class PeriodFn, class OutFn>
    RFS_HD void period_over(const PeriodFn& T, const OutFn& out, bool failed) {
        for (;;) {
            int go = 0;                                  // 1: next mode, 2: next retry period
            if (failed) {
                if (!retry) {
                    if (iq == 0) fund_failed = 1;        // only the fundamental sets ierr
                    ift = k;
                    for (int i = k; i < kmax; i++) out(i, 0.0);
                    go = 1;
                } else {
                    if (iq == 0) { flag = 0; done = 1; return; }       // the single-period call returns ierr = 1
                    out(k, 0.0);                         // (every later mode of that call is capped at this period)
                    go = 2;
                }
            } else if (retry) {
                iq++;
                if (iq < nmode) { start_period(T); return; }
                go = 2;
            } else {
                k++;
                if (k >= kmax) go = 1;
                else if (k >= ift) { failed = true; continue; }        // :236 `if(k.ge.ift) go to 1700`
                else { start_period(T); return; }
            }
            if (go == 1) {
                iq++;
                if (iq < nmode) {
                    k = 0;
                    if (k >= ift) { failed = true; continue; }
                    start_period(T); return;
                }
                if (!fund_failed) { done = 1; return; }
                retry = 1; k = -1;
            }
            // retry pass: the next period whose output is zero / NaN, all modes again, the period on its own
            for (k = k + 1; k < kmax; k++) { const double v = out.get(k); if (v == 0.0 || v != v) break; }
            if (k >= kmax) { done = 1; return; }
            iq = 0; ift = 1 << 30;
            start_period(T); return;
        }
    }

    RFS_HD void set_modes(int nmode_, double* cmb_, long cms_) { nmode = nmode_; cmb = cmb_; cms = cms_; }

    template <class PeriodFn>
    RFS_HD void begin(const SwdModel& M, const PeriodFn& T, int kmax_) {
        // (the P velocity only enters through the water-layer test and the half-space start value gtsolh)
        // surfdisp96.f:149-160 extremal velocities, :203-222 start value
        float bmx; float cc1 = swd_start_value(M, bmx);
        cc = (double)cc1; dc = (double)0.005f; cm = cc; betmx = bmx;
        kmax = kmax_; k = 0; retry = 0; done = 0; flag = 1; nsec = 0;
        iq = 0; ift = 1 << 30; fund_failed = 0;
        del1st = 0.0; cprev = 0.0; m = 1; nev = 1; nctrl = 1;
        if (!Tab::kDynamic) for (int i = 0; i < 12; i++) { tab.sx(i, 0.0); tab.sy(i, 0.0); }
        if (kmax <= 0) { done = 1; return; }
        // (a model whose fastest layer lies SWD_MAX_SCAN above the start value -- velocities that are not velocities, or not
        // numbers: a position that left its bounds, a caller's mistake -- would have the scan walk millions of cells of 0.005,
        // minutes on a device whose other chains wait for this one; the reference would walk them.  Such a search fails.)
        absurd = !((double)betmx - cc < SWD_MAX_SCAN) ? 1 : 0;
        start_period(T);
    }

    // Where the scan would be j steps from now if no sign change came (the `cont` path of advance() and its next-request
    // block, repeated): false when the pending request is not a scan point or a limit would end the scan first.  Only
    // an efficiency device -- a caller evaluates the point ahead of time and offers the result to advance() once the
    // machine really asks for that very point.
    RFS_HD bool scan_peek(int j, double& c) const {
        if (phase != PH_SCAN) return false;
        double a1 = c1, a2 = c2; int id = idir;
        for (int i = 0; i < j; i++) {
            a1 = a2;
            if (!(a1 >= cm && a1 < ((double)betmx + dc))) return false;
            const double c2n = (id > 0) ? a1 + dc : a1 - dc;
            const bool clamp = c2n <= clow;
            if (clamp) { id = +1; a1 = clow; }
            a2 = clamp ? a1 + dc : c2n;
        }
        c = a2;
        return true;
    }

    // consume del = secular(creq) and run until the next request (or completion).
    // Forward-only staging (phase dispatch -> LOOPTOP -> A1 -> FINISH -> FAIL -> SCAN / new period): every
    // transition of getsol / nevill moves forward through these stages, so there is no loop over states and
    // the wavefront executes each stage at most once per call whatever mixture of phases its lanes are in.
    template <class PeriodFn, class OutFn>
    RFS_HD void advance(double del, const PeriodFn& T, const OutFn& out) {
        nsec++;
        if (absurd) { for (int kk = 0; kk < kmax; kk++) out(kk, 0.0); flag = 0; done = 1; return; }
        // Phase dispatch and the loop head of nevill as PREDICATED updates (selects, no branches): with 64 lanes in a
        // mixture of phases every branch of an if / else chain would be executed anyway, and the structurised control
        // flow cost several times the dozen selects below.  sgn1(a) * sgn1(b) < 0  <=>  the sign bits differ.
        const int ph = phase;
        const bool pS = ph == PH_START, pC = ph == PH_SCAN, pH0 = ph == PH_HALF0, pHO = ph == PH_HALF_OUT,
                   pHB = ph == PH_HALF_B, pN = ph == PH_NEV;
        // getsol head (surfdisp96.f:433-447) and one scan step (:470-479)
        del1st = (pS && ifirst == 1) ? del : del1st;
        idir = pS ? (diffsign(del1st, del) ? -1 : +1) : idir;              // ifirst == 1: del1st == del -> +1
        const bool chg = pC && diffsign(del1, del);                         // bracket found
        const bool cont = pC && !chg;
        del2 = pC ? del : del2;
        c1 = cont ? c2 : c1;
        del1 = (pS || cont) ? del : del1;
        // (the limits as the reference states them, getsol :477-479, negated so that a NaN scan point -- a model with NaN
        // velocities: a sampler that carried a NaN gradient into its next drift -- ends the scan as well instead of walking on for ever)
        const bool sfail = cont && !(c1 >= cm && c1 < ((double)betmx + dc));
        const bool st_scan = pS || (cont && !sfail);
        bool st_fail = sfail, st_half = chg, st_newperiod = false;
        bool return_after = false;                                          // (MODES: the next request is already set up)
        int half_phase = PH_HALF0;
        // entries of nevill's loop (:590-594 and the three ways back to its top)
        del3 = (pH0 || pHO || pHB || pN) ? del : del3;
        nev = (pH0 || pHB) ? 1 : (pN ? 2 : nev);
        m = pHB ? 1 : (pN ? (m >= 10 ? 10 : m + 1) : m);
        const bool st_looptop = pH0 || pHB || pN;
        nctrl = pH0 ? 2 : (st_looptop ? nctrl + 1 : nctrl);                 // :595 (HALF0 starts from nctrl = 1)
        const bool lt_fin = st_looptop && nctrl >= 100;
        const bool lt_out = st_looptop && !lt_fin && (c3 < fmin(c1, c2) || c3 > fmax(c1, c2));   // :597-607
        nev = lt_out ? 0 : nev;
        st_half = st_half || lt_out;
        half_phase = lt_out ? (int)PH_HALF_OUT : half_phase;
        bool st_finish = lt_fin;
        const bool st_a1 = pHO || (st_looptop && !lt_fin && !lt_out);
        if (st_a1) {                                     // nevill :608-681
            const double s13 = del1 - del3, s32 = del3 - del2;
            const bool opp = diffsign(del3, del1);
            c2 = opp ? c3 : c2; del2 = opp ? del3 : del2;
            c1 = opp ? c1 : c3; del1 = opp ? del1 : del3;
            if (fabs(c1 - c2) <= 1.0e-6 * c1) st_finish = true;
            else {
                if (diffsign(s13, s32)) nev = 0;
                const double pct = (double)0.01f;        // default-real literal 0.01 (:637,639)
                double ss1 = fabs(del1), s1 = pct * ss1, ss2 = fabs(del2), s2 = pct * ss2;
                if (s1 > ss2 || s2 > ss1 || nev == 0) { st_half = true; half_phase = PH_HALF_B; }
                else {
                    double ym1;
                    bool bail = false;
                    double xn = 0.0;                     // x(j+1) of the running recurrence (kDynamic form)
                    if (Tab::kDynamic) {
                        // x(j+1) travels in a register and the operands of step j-1 are fetched before step j's
                        // division: the memory round trip of the table leaves the dependent chain (same arithmetic)
                        if (nev == 2) { tab.sx(m + 1, c3); tab.sy(m + 1, del3); ym1 = del3; xn = c3; }
                        else { tab.sx(1, c1); tab.sy(1, del1); tab.sx(2, c2); tab.sy(2, del2); m = 1; ym1 = del2; xn = c2; }
                        double yj = tab.gy(m), xj = tab.gx(m);
                        for (int j = m; j >= 1 && !bail; j--) {
                            double yn = 0.0, xq = 0.0;
                            if (j > 1) { yn = tab.gy(j - 1); xq = tab.gx(j - 1); }
                            double denom = ym1 - yj;
                            if (fabs(denom) < 1.0e-10 * fabs(ym1)) bail = true;
                            else { xn = (-yj * xn + ym1 * xj) / denom; tab.sx(j, xn); }
                            yj = yn; xj = xq;
                        }
                    } else {
                        if (nev == 2) {
#pragma unroll
                            for (int i = 2; i <= 11; i++) if (i == m + 1) { tab.sx(i, c3); tab.sy(i, del3); }
                            ym1 = del3;
                        } else {
                            tab.sx(1, c1); tab.sy(1, del1); tab.sx(2, c2); tab.sy(2, del2); m = 1; ym1 = del2;
                        }
#pragma unroll
                        for (int j = 10; j >= 1; j--) {
                            if (j <= m && !bail) {
                                double denom = ym1 - tab.gy(j);
                                if (fabs(denom) < 1.0e-10 * fabs(ym1)) bail = true;
                                else tab.sx(j, (-tab.gy(j) * tab.gx(j + 1) + ym1 * tab.gx(j)) / denom);
                            }
                        }
                    }
                    if (bail) { st_half = true; half_phase = PH_HALF_B; }
                    else { c3 = Tab::kDynamic ? xn : tab.gx(1); creq = c3; phase = PH_NEV; }
                }
            }
        }
        if (st_finish) {                                 // getsol :483-487
            c1 = c3;
            if (c1 > (double)betmx) st_fail = true;
            else {
                out(k, (double)(float)c1);
                cprev = c1;
                if constexpr (MODES) {
                    cmb[(long)k * cms] = c1;             // c(k) = c1 (:277), unrounded: the next mode's floor
                    period_over(T, out, false); return_after = true;
                } else {
                    k = k + 1;
                    if (k >= kmax) done = 1; else st_newperiod = true;
                }
            }
        }
        if (st_fail) {
            if constexpr (MODES) { period_over(T, out, true); return_after = true; }
            else if (!retry) {                           // surfdisp96.f:317-362 + surfdisp.cpp:93-100
                retry = 1;
                for (int i = k; i < kmax; i++) out(i, 0.0);
                st_newperiod = true;
            } else { flag = 0; done = 1; }
        }
        if (MODES && return_after) return;
        {   // next request of the lanes that stay inside the period: a bisection point or the next scan point
            // (getsol loop 1000, :457-469); predicated, the two cases exclude each other
            const double c2n = (idir > 0) ? c1 + dc : c1 - dc;
            const bool clamp = st_scan && c2n <= clow;                      // del1 kept (quirk)
            idir = clamp ? +1 : idir;
            c1 = clamp ? clow : c1;
            c2 = st_scan ? (clamp ? c1 + dc : c2n) : c2;
            c3 = st_half ? 0.5 * (c1 + c2) : c3;
            creq = st_half ? c3 : (st_scan ? c2 : creq);
            phase = st_half ? half_phase : (st_scan ? (int)PH_SCAN : phase);
        }
        if (st_newperiod) start_period(T);
    }
};
using RootSearch = RootSearchT<NevTabReg>;
using RootSearchModes = RootSearchT<NevTabReg, true>;

// ---------------------------------------------------------------------------
// Warm-started root refinement for the leapfrog loop (no counterpart in the reference, which searches every model
// from scratch: surfdisp96.f:257-316).  Inside a trajectory the model of step s differs from that of step s-1 by
// dx = dt M^-1 p, and step s-1 left both its roots c_k and their Frechet kernels dc_k/dm behind.  Every (period,
// chain) item is therefore independent of the other periods:
//     predict   c_pred = c_prev + sum_j G_j dx_j          (first order; G = the kernels through the chain rule)
//     bracket   [c_pred - eps, c_pred + eps], widened x4 (likelier side first) while the secular function keeps its
//               sign, never beyond the trust radius R = R0 c + R1 sum_j |G_j dx_j|
//     refine    false position with the Illinois rule until two successive estimates agree to WARM_TOL c
// about 3 secular evaluations instead of the ~23 of the sequential scan + nevill, and lane = (period, chain).
// From the second continued step on the item also has the SLOPE of the secular function at its previous root: the search
// then starts with a Newton step from c_pred, overshot by a quarter, so that two evaluations bracket the root within a
// fraction of the prediction error; where that bracket is narrow and its slope agrees with the one the step was taken
// with (the function is linear across it), its secant point is the root: 2 evaluations.  Anything else falls back to the
// bracket above.
// A second, one-evaluation test keeps the continued root on the branch the reference's scan would pick: the scan of
// period k starts at c(k-1) - 1.5 dc (the first period at the start value of the model, surfdisp96.f:257-276) and takes
// the first sign change it meets.  Where the dispersion is normal at every period of the sequence (each root above its
// period's start point) one evaluation at the start point does: the function there must have the sign it has just
// below the continued root, otherwise another root has moved in between.  A sequence with anomalous dispersion somewhere
// (velocity inversions: crowded spectra, and the reference's pick among neighbouring modes then depends on where its
// 0.005 km/s grid falls) walks the reference's own scan grid for every period, up or down as getsol would, and the first
// cell with a sign change must hold the continued root (k_swd_warm_check).  Anything else: back to the full search.
// The machine only REQUESTS evaluations, like RootSearchT.  It declines (status W_FAIL) whenever anything is off --
// no sign change inside the trust radius, no convergence, a root above the fastest layer -- and the caller then runs
// the reference-semantics search for that chain, which alone decides flags.  Accepted roots lie within WARM_TOL c of
// a sign change of the very function the reference search brackets, i.e. inside the reference's own refinement
// tolerance 1e-6 c (surfdisp96.f:627); they are rounded to float32 like the reference's (surfdisp96.f:302).
// ---------------------------------------------------------------------------
constexpr double WARM_TOL = 1.0e-7;          // relative agreement of two successive estimates
constexpr double WARM_EPS0 = 1.0e-5;         // first bracket half-width, relative to c (+ WARM_EPS1 * l1)
constexpr double WARM_EPS1 = 0.03;
constexpr double WARM_R0 = 2.0e-4;           // trust radius: R0 c + R1 l1
constexpr double WARM_R1 = 1.0;              // (a first-order model that misses by more than its own size is no guide)
constexpr double WARM_L1MAX = 0.5;           // km/s: beyond this first-order change the one-evaluation branch test is not trusted:
                                             // the chain's sequences walk the reference's scan grid (k_swd_warm_walk) like irregular ones
constexpr double WARM_L1WIDE = 2.0;          // km/s: and beyond this the model is not "the previous one, moved" at all
constexpr int WARM_MAXIT = 12;
// Round 5: beyond the trust radius.  Where no sign change lies within R of the first-order prediction (the root has left the
// first-order model's reach, or two modes have come close: 32 of the 53 chains a bench step handed back), the bracket keeps
// widening up to WARM_RWIDE x R, at most WARM_RWIDE_ABS km/s -- whatever root that finds, the first-order model does not vouch
// for it, so the lane reports `wide` and every sequence of the chain walks the reference's scan grid (k_swd_warm_walk): the
// walk alone decides whether the reference's scan would stop in that root's cell.
constexpr double WARM_RWIDE = 16.0;
constexpr double WARM_RWIDE_ABS = 0.1;
// ... and a bracket that false position has not closed in WARM_MAXIT steps (the normalised secular function is a step
// between crowded modes: +-1 either side) is closed by bisection
constexpr int WARM_MAXBIS = 48;


constexpr double WARM_OVER = 0.25;           // Newton start: overshoot of the step, so that the second point lands beyond the root
constexpr double WARM_W2 = 3.0e-5;           // ... and a bracket this narrow (relative to c) whose slope agrees with the slope the step
constexpr double WARM_LIN = 0.02;            // was taken with to WARM_LIN is settled by its secant point: the secant's error is
                                             // (slope change across the bracket) x w / 8 <= 0.02 x 3e-5 c / 8 < 1e-7 c; between crowded
                                             // modes the slopes disagree and the third evaluation is made

struct WarmSearch {
    enum { W_A, W_B, W_X, W_REF, W_N0, W_N1, W_DONE, W_FAIL };
    double cpred, eps, R, a, fa, b, fb, creq, root, slope, f0, mlast;
    int phase, it, side, second, lastside, nev, ntry;
    // the bracket was found beyond the trust radius: the caller makes the chain's sequences walk the grid
    // (no state of its own -- the kernel sits at the edge of three wavefronts per SIMD: eps only ever exceeds R out there)
    RFS_HD bool wide() const { return eps > R; }

    RFS_HD bool active() const { return phase < W_DONE; }

    // The machine's small integers (+ the caller's attempt flag and the evaluations of its first attempt) in one word, and
    // back: a search that moves to another lane between two evaluations (k_swd_warm's rounds) carries its 12 doubles and this.
    RFS_HD unsigned long long pack_small(int attempt, int nev_first) const {
        return (unsigned long long)(phase & 7) | ((unsigned long long)(it & 127) << 3) | ((unsigned long long)(side & 1) << 10) |
               ((unsigned long long)(second & 1) << 11) | ((unsigned long long)((lastside + 1) & 3) << 12) |
               ((unsigned long long)(ntry & 3) << 14) | ((unsigned long long)(attempt & 1) << 16) |
               ((unsigned long long)(nev & 4095) << 20) | ((unsigned long long)(nev_first & 4095) << 32);
    }
    RFS_HD void unpack_small(unsigned long long bt, int& attempt, int& nev_first) {
        phase = (int)(bt & 7); it = (int)((bt >> 3) & 127); side = (int)((bt >> 10) & 1); second = (int)((bt >> 11) & 1);
        lastside = (int)((bt >> 12) & 3) - 1; ntry = (int)((bt >> 14) & 3); attempt = (int)((bt >> 16) & 1);
        nev = (int)((bt >> 20) & 4095); nev_first = (int)((bt >> 32) & 4095);
    }

    // cprev: root of the previous model; dc: first-order change; l1: sum of |first-order terms|; slope0: d(secular)/dc at
    // the previous model's root as its own search left it (0 = unknown).  With a slope the search starts with a Newton
    // step from the prediction, overshot by WARM_OVER: two evaluations bracket the root within a fraction of the
    // prediction error, and a bracket that narrow needs no third.
    RFS_HD void begin(double cprev, double dc, double l1, double slope0 = 0.0) {
        cpred = cprev + dc;
        R = WARM_R0 * cpred + WARM_R1 * l1;
        nev = 0; it = 0; side = 0; second = 0; lastside = -1; root = 0.0; slope = 0.0; f0 = 0.0; ntry = 0; mlast = 0.0;
        fa = fb = 0.0;
        eps = WARM_EPS0 * cpred + WARM_EPS1 * l1;
        a = cpred - eps; b = cpred + eps; creq = a;
        phase = W_A;
        // not a continuation of the previous model (or no previous root at all): leave it to the full search
        if (!(cprev > 0.0) || !(l1 <= WARM_L1WIDE) || !(a > 0.0)) phase = W_FAIL;
        // ... unless the search may go beyond the trust radius anyway (begin_wide below)
        else if (slope0 != 0.0 && slope0 == slope0 && l1 <= WARM_L1MAX) { slope = slope0; creq = cpred; phase = W_N0; }
    }

    // A first-order change beyond WARM_L1WIDE (kernels blown up next to an osculation point: 11 chains of a bench step) says
    // nothing about where the root went -- but the root itself has usually moved little.  With the wide search allowed, such
    // an item looks around the PREVIOUS root instead, out to WARM_RWIDE_ABS; the caller marks the chain wide (its l1 is
    // beyond WARM_L1MAX), so the grid walk decides whether what is found is the reference's root.
    RFS_HD void begin_wide(double cprev) {
        const double l1w = WARM_RWIDE_ABS / WARM_RWIDE;
        cpred = cprev; R = WARM_R0 * cpred + WARM_R1 * l1w; eps = WARM_EPS0 * cpred + WARM_EPS1 * l1w;
        slope = 0.0;
        start_bracket();
        if (!(cprev > 0.0) || !(a > 0.0)) phase = W_FAIL;
    }

    RFS_HD void start_bracket() { a = cpred - eps; b = cpred + eps; creq = a; phase = W_A; }

    RFS_HD void refine_from_bracket() {          // (a, fa), (b, fb) hold a sign change
        slope = (fb - fa) / (b - a);
        const double c3 = a - fa * (b - a) / (fb - fa);
        creq = c3; phase = W_REF; lastside = -1;
    }

    // may_widen: allowed to look beyond the trust radius at all (rfs_set_option "swd_warm_widen")
    RFS_HD void advance(double f, const bool may_widen = true) {
        nev++;
        if (f != f) { phase = W_FAIL; return; }
        bool widen = false;
        if (phase == W_N0 || phase == W_N1) {
            // (a, f0): the point before this one (W_N1); creq: the point just evaluated
            if (phase == W_N1 && diffsign(f0, f)) {                 // the two points bracket the root
                const double c0 = a, c1 = creq;
                if (c0 < c1) { a = c0; fa = f0; b = c1; fb = f; } else { a = c1; fa = f; b = c0; fb = f0; }
                const double mb = (fb - fa) / (b - a);
                if (b - a <= WARM_W2 * b && fabs(mb - mlast) <= WARM_LIN * fabs(mb)) { slope = mb; root = a - fa * (b - a) / (fb - fa); phase = W_DONE; }
                else refine_from_bracket();
                return;
            }
            // the first point, or a second one on the same side: step on with the best slope at hand
            const double m = (phase == W_N1) ? (f - f0) / (creq - a) : slope;
            const double step = -(1.0 + WARM_OVER) * f / m;
            if (ntry >= 2 || !(fabs(step) <= R) || step == 0.0 || !(fabs(creq + step - cpred) <= R)) { start_bracket(); return; }
            a = creq; f0 = f; ntry++; mlast = m;
            creq = creq + step; phase = (creq > 0.0) ? (int)W_N1 : (int)W_FAIL;
            return;
        }
        if (phase == W_A) { fa = f; creq = b; phase = W_B; }
        else if (phase == W_B) {
            fb = f;
            if (diffsign(fa, fb)) refine_from_bracket();
            else { second = 0; widen = true; }
        } else if (phase == W_X) {                 // a point further out on `side` was evaluated
            bool found = false;
            if (side == 0) {
                if (diffsign(f, fa)) { b = a; fb = fa; a = creq; fa = f; found = true; }
                else { a = creq; fa = f; }
            } else {
                if (diffsign(f, fb)) { a = b; fa = fb; b = creq; fb = f; found = true; }
                else { b = creq; fb = f; }
            }
            if (found) refine_from_bracket();
            else if (!second) {                    // the other side at the same distance
                second = 1; side = 1 - side;
                creq = side == 0 ? cpred - eps : cpred + eps;
                if (!(creq > 0.0)) phase = W_FAIL;
            } else { second = 0; widen = true; }
        } else if (phase == W_REF) {
            const double c3 = creq;
            if (f == 0.0) { root = c3; phase = W_DONE; return; }
            if (!diffsign(f, fa)) { a = c3; fa = f; if (lastside == 0) fb *= 0.5; lastside = 0; }    // Illinois rule
            else { b = c3; fb = f; if (lastside == 1) fa *= 0.5; lastside = 1; }
            const double c4 = a - fa * (b - a) / (fb - fa);
            if (it >= WARM_MAXIT) {                // bisection from here on (the signs of fa / fb are all that is used)
                if (b - a <= 2.0 * WARM_TOL * b) { root = 0.5 * (a + b); phase = W_DONE; }
                else if (++it >= WARM_MAXIT + WARM_MAXBIS) phase = W_FAIL;
                else creq = 0.5 * (a + b);
            }
            else if (fabs(c4 - c3) <= WARM_TOL * fabs(c4)) { root = c4; phase = W_DONE; }
            else if (!(c4 >= a && c4 <= b)) phase = W_FAIL;
            else if (++it >= WARM_MAXIT) creq = 0.5 * (a + b);
            else creq = c4;
        }
        if (widen) {
            const double Rw = may_widen ? fmin(WARM_RWIDE * R, fmax(R, WARM_RWIDE_ABS)) : R;
            if (eps >= Rw) { phase = W_FAIL; return; }
            eps = eps < R ? fmin(4.0 * eps, R) : fmin(4.0 * eps, Rw);      // (beyond R: not on the first-order model's word any more, wide())
            side = (fabs(fa) <= fabs(fb)) ? 0 : 1;           // the side the function is closer to zero on goes first
            creq = side == 0 ? cpred - eps : cpred + eps;
            phase = (creq > 0.0) ? (int)W_X : (int)W_FAIL;
        }
    }
};

// ---------------------------------------------------------------------------
// The reference's own roots, period-parallel (round 4).  A warm-started root is the converged sign change; the reference's
// is what its nevill returns: the last midpoint of a run of bisection steps, 0.5 .. 1.0e-6 c short of the sign change
// (surfdisp96.f:568-687) -- a function of the scan cell [c1, c1 +- dc] the bracket was found in, i.e. of the grid the scan of
// that period walks: origin = the UNROUNDED root of the period before - 1.5 dc (:272-275), or the model's start value for a
// sequence's first period (:257-260).  Given the right cell, nevill needs nothing else from the scan, so a lane can
// produce the reference's root of period k bit for bit from (origin, approximate root): step the grid from the origin to
// the cell that holds the approximate root (the same repeated additions as getsol's loop), evaluate the two edges, run the
// reference's refinement (RootSearchT::advance from the state getsol hands it) inside.  The chain through the origins is
// sequential, but strongly contracting: an origin that is off by delta moves the result by ~delta / 2^j (j ~ 10 bisections)
// unless a decision of nevill flips (measured with delta = 1e-6 c: median 1.2e-9 c, 5 % of the float32 results differ;
// with the exact origin: identical, 2 496 of 2 496).  So the periods of a sequence are cut into GROUPS; a lane takes one
// group, sequentially, handing the unrounded root from period to period exactly as the reference does, and precedes it
// with `runup` periods whose only purpose is the origin of the group's first period: the first run-up period takes its
// origin from the warm root of the period before it (1e-6 c off), the next ones inherit errors of 1e-9 c, 1e-12 c, ...
// The first group of a sequence needs none: its first origin is the model's start value.
// The machine only REQUESTS evaluations, like RootSearchT and WarmSearch.  It declines (status X_FAIL) whatever does not
// fit -- no sign change in the cell the approximate root lies in nor in its neighbour, a grid that would touch the lower
// clamp of getsol (:463-467), a root the reference rejects (:483-485) -- and the caller hands the chain to the
// reference-semantics search.
// ---------------------------------------------------------------------------
constexpr double EXACT_ORIGIN_ERR = 4.0e-7;    // ... which leaves the first run-up origin within this of the reference's (relative)
constexpr double EXACT_ORIGIN_TOL = 1.0e-7;    // origin accuracy a wanted period needs (see ExactGroupT::step_nevill): two run-up periods of a smooth secular function leave ~1e-11, roots found by bisections alone leave the 4e-7
constexpr int EXACT_SPILL_ND = 68;           // doubles of ExactGroupT::save / load
constexpr double EXACT_OFFSET = 0.75e-6;     // the reference's root lies 0.5 .. 1.0e-6 c below the sign change: the run-up's first origin is moved by the mean

// nevill (surfdisp96.f:568-687) inside a bracket, as a request machine of its own -- the same decisions and the same
// arithmetic as RootSearchT::advance's refinement stages (a CPU test of the lane code holds the two together bit for
// bit) -- with LAZY evaluation: of the ~12 points nevill evaluates, most are midpoints of a run of bisections that
// approaches the root from the FAR end of a bracket whose other end already sits on the root (|f| a few 1e-6 of the far
// value after the first interpolation, 1e-10 after the second).  All nevill does with such a value is (a) take its sign
// -- the far end's --, (b) test whether it is still more than 100 times the near end's (:637-641: then it bisects again)
// and (c) hand it to the next test of the same kind.  Where the function is close to linear across the bracket -- judged
// from the residual of the first interpolation, which is exactly its curvature -- the value is the mean of the two ends to
// a fraction of a per cent, so the machine supplies that mean itself as long as it exceeds the near value by more than
// 100 x LAZY_MARGIN, and asks for the true value where the decision is closer, where the point will enter an
// interpolation (interpolated points and the ends they are built from are always true values), and whenever the bracket
// does not look like that.  The points themselves -- midpoints of exact doubles -- do not depend on the values, so the
// returned root is the reference's bit for bit while 7-8 of its ~15 evaluations are never made.
constexpr double LAZY_MARGIN = 1.5;          // a supplied value decides only tests it wins by this factor (its error where it is used: < 2 %)
constexpr double LAZY_LIN = 0.02;            // residual of the first interpolation relative to the far end: above this, no supplied values

template <bool LAZY = true>
struct CellNevillT {
    enum { N_HALF0, N_HALF_OUT, N_HALF_B, N_NEV, N_FIX1, N_FIX2, N_DONE, N_FAIL };
    double c1, c2, del1, del2, c3, del3v, creq, result;
    NevTabMem tab;                           // Neville table x(1..11), y(1..11) (1-based like the Fortran): LDS on the device (set by the owner)
    int phase, nev, m, nctrl, nsupplied;
    bool ex1, ex2, lin;
    float betmx;
    // How an error of the cell's position (the scan's origin: the root of the period before) reaches the points: the cell's
    // edges move with it one to one, a midpoint by the mean of its two ends, an interpolated point hardly at all (the root
    // of a smooth function does not care where it is interpolated from).  g3 at the end = the factor by which the returned
    // root inherits the origin's error: ~2^-j after j closing bisections towards an interpolated end (1e-3 typically), 1 for
    // a root found by bisections alone (a step-like function: crowded spectra).
    float g1, g2, g3;

    RFS_HD bool active() const { return phase < N_DONE; }

    // (c1, del1): the scan's last point before the sign change, (c2, del2) its first point behind it; both values true
    RFS_HD void enter(double c1_, double del1_, double c2_, double del2_, float bmx) {
        c1 = c1_; del1 = del1_; c2 = c2_; del2 = del2_; betmx = bmx;
        ex1 = ex2 = true; lin = false; nev = 1; m = 1; nctrl = 1; nsupplied = 0; result = 0.0;
        g1 = g2 = 1.0f; g3 = 1.0f;
        c3 = 0.5 * (c1 + c2); creq = c3; phase = N_HALF0;                 // nevill :583-589 (half)
    }

    RFS_HD void finish() {                                                 // getsol :483-487
        result = c3;
        phase = (c3 > (double)betmx) ? (int)N_FAIL : (int)N_DONE;
    }

    // the part of nevill's loop body behind the bracket update: ratio tests, Neville step or bisection (:636-681)
    RFS_HD void choose() {
        const double pct = (double)0.01f;
        const double ss1 = fabs(del1), ss2 = fabs(del2);
        bool half = (pct * ss1 > ss2) || (pct * ss2 > ss1) || nev == 0;
        if (!half) {
            // an interpolation: every value it uses must be a true one
            if (nev != 2 && !(ex1 && ex2)) { creq = ex1 ? c2 : c1; phase = ex1 ? (int)N_FIX2 : (int)N_FIX1; return; }
            double xn;
            if (nev != 2) {
                // table rebuilt from the bracket's ends (m = 1): one step of the recurrence
                m = 1;
                const double denom = del2 - del1;
                half = fabs(denom) < 1.0e-10 * fabs(del2);
                xn = (-del1 * c2 + del2 * c1) / denom;
                tab.sx(1, half ? c1 : xn); tab.sy(1, del1); tab.sx(2, c2); tab.sy(2, del2);
            } else {
                // one more point on the table (rare: two interpolations in a row)
                tab.sx(m + 1, c3); tab.sy(m + 1, del3v);
                const double ym1 = del3v;
                xn = c3;
                for (int j = m; j >= 1 && !half; j--) {
                    const double yj = tab.gy(j), denom = ym1 - yj;
                    if (fabs(denom) < 1.0e-10 * fabs(ym1)) half = true;
                    else { xn = (-yj * xn + ym1 * tab.gx(j)) / denom; tab.sx(j, xn); }
                }
            }
            if (!half) { c3 = xn; creq = c3; phase = N_NEV; return; }      // (g3: once its value is known, step())
        }
        c3 = 0.5 * (c1 + c2); creq = c3; phase = N_HALF_B; g3 = 0.5f * (g1 + g2);
    }

    // one pass of nevill's loop with the TRUE value del3 of f(c3)
    RFS_HD void step(double del3) {
        const int ph = phase;
        const bool looptop = ph != N_HALF_OUT;
        nev = (ph == N_HALF0 || ph == N_HALF_B) ? 1 : (ph == N_NEV ? 2 : nev);
        m = (ph == N_HALF_B) ? 1 : (ph == N_NEV ? (m >= 10 ? 10 : m + 1) : m);
        nctrl = (ph == N_HALF0) ? 2 : (looptop ? nctrl + 1 : nctrl);
        if (looptop) {
            if (nctrl >= 100) { finish(); return; }                                  // :595
            if (c3 < fmin(c1, c2) || c3 > fmax(c1, c2)) {                             // :597-607
                nev = 0; c3 = 0.5 * (c1 + c2); creq = c3; phase = N_HALF_OUT; g3 = 0.5f * (g1 + g2); return;
            }
        }
        if (ph == N_NEV) {
            // the residual of an interpolated point against the ends it was built from: the nonlinearity of the function
            // across the bracket (it would be zero).  The first one decides about supplied values (LAZY); every one says how
            // far the point follows the ends when they move (g3): not at all where the function is linear, like a mean of
            // the two where it is a step (both ends saturated: crowded spectra)
            const double q = fabs(del3) / fmax(fmax(fabs(del1), fabs(del2)), 1.0e-300);
            if (LAZY && nctrl == 3) lin = q <= LAZY_LIN;
            g3 = fminf(1.0f, 4.0f * (float)q) * fmaxf(g1, g2);
        }
        const double s13 = del1 - del3, s32 = del3 - del2;
        const bool opp = diffsign(del3, del1);                                        // :608-617
        c2 = opp ? c3 : c2; del2 = opp ? del3 : del2; ex2 = opp ? true : ex2; g2 = opp ? g3 : g2;
        c1 = opp ? c1 : c3; del1 = opp ? del1 : del3; ex1 = opp ? ex1 : true; g1 = opp ? g1 : g3;
        if (fabs(c1 - c2) <= 1.0e-6 * c1) { finish(); return; }                      // :627
        if (diffsign(s13, s32)) nev = 0;
        del3v = del3;
        choose();
    }

    // may the value at the pending bisection point be supplied?  (see the struct's comment)
    RFS_HD bool can_supply(double& v) const {
        if (!LAZY || !lin || phase != N_HALF_B) return false;
        const double a1 = fabs(del1), a2 = fabs(del2);
        const bool near1 = a1 < a2;
        if (near1 ? !ex1 : !ex2) return false;                       // the near end's value is a true one
        v = 0.5 * (del1 + del2);
        return fabs(v) > (100.0 * LAZY_MARGIN) * (near1 ? a1 : a2);   // (also: the sign of v is the far end's by far)
    }

    // A supplied value at a far-side bisection point: what step() does with it, and nothing else -- the point is the
    // bracket's midpoint (inside by construction), the value lies between the ends' (no monotonicity flag, :630-634), it
    // has the far end's sign and replaces that end, and the ratio test that follows is the one can_supply() has decided
    // (bisect again) unless the bracket has become narrow enough to stop.
    RFS_HD void supplied_step(double v) {
        nev = 1; m = 1; nctrl++;
        if (nctrl >= 100) { finish(); return; }
        const bool near1 = fabs(del1) < fabs(del2);
        c2 = near1 ? c3 : c2; del2 = near1 ? v : del2; ex2 = near1 ? false : ex2; g2 = near1 ? g3 : g2;
        c1 = near1 ? c1 : c3; del1 = near1 ? del1 : v; ex1 = near1 ? ex1 : false; g1 = near1 ? g1 : g3;
        if (fabs(c1 - c2) <= 1.0e-6 * c1) { finish(); return; }
        c3 = 0.5 * (c1 + c2); creq = c3; g3 = 0.5f * (g1 + g2);            // (phase stays N_HALF_B)
    }

    // consume the TRUE value f of f(creq); afterwards a new request is pending or the machine is done / has failed
    RFS_HD void advance(double f) {
        if (phase == N_FIX1 || phase == N_FIX2) {
            if (phase == N_FIX1) { del1 = f; ex1 = true; } else { del2 = f; ex2 = true; }
            choose();
        } else step(f);
        double v;
        while (can_supply(v)) { nsupplied++; supplied_step(v); }
    }
};
using CellNevill = CellNevillT<true>;

template <bool LAZY = true>
struct ExactGroupT {
    enum { X_E1, X_E2, X_E1B, X_NEV, X_DONE, X_FAIL };
    CellNevillT<LAZY> nv;
    double creq, omega;                      // the pending request: secular function of period `k` at creq
    double o, rhat, c1, c2, del1, del2s, cprev, cc, dcs;
    int k, k0, k1, phase, dir, msteps, shifted, nev, cause, nsupplied;
    float betmx;
    float oerr;                              // relative error of the current origin, as far as the factors g3 of the periods so far tell
    float otol;                              // ... and what a wanted period accepts (EXACT_ORIGIN_TOL; rfs_set_option "swd_exact_origin_tol_e9")

    RFS_HD bool active() const { return phase < X_DONE; }

    // grid point number m of the scan from origin o (getsol's own repeated additions, surfdisp96.f:457-469)
    RFS_HD double grid(int m) const {
        double c = o;
        for (int i = 0; i < m; i++) c = (dir > 0) ? c + dcs : c - dcs;
        return c;
    }

    // Periods [kr, k1) of one sequence, results wanted for [k0, k1) (kr < k0: run-up).  cstart: the model's start value
    // (swd_start_value), bmx: its fastest S velocity.  origin0: unrounded root of period kr - 1 as far as it is known
    // (ignored for kr == 0).
    // tab: storage of the Neville table, 24 doubles at stride `tabstride` (LDS on the device: one column per lane)
    template <class RootFn, class OmegaFn>
    RFS_HD void begin(int kr, int k0_, int k1_, double cstart, float bmx, double origin0, const RootFn& approx, const OmegaFn& om,
                      double* tab, int tabstride, float origin_tol = (float)EXACT_ORIGIN_TOL) {
        otol = origin_tol;
        k = kr; k0 = k0_; k1 = k1_; cc = cstart; betmx = bmx; dcs = (double)0.005f;
        cprev = origin0; nev = 0; cause = 0; nsupplied = 0;
        oerr = kr > 0 ? (float)EXACT_ORIGIN_ERR : 0.0f;                  // (a sequence's first period starts at the model's start value: exact)
        nv.tab.base = tab; nv.tab.stride = tabstride;
        start_period(approx, om);
    }

    template <class RootFn, class OmegaFn>
    RFS_HD void start_period(const RootFn& approx, const OmegaFn& om) {
        rhat = approx(k); omega = om(k);
        o = (k == 0) ? cc : cprev - 1.5 * dcs;                       // surfdisp96.f:257-260 / :272-275
        if (!(rhat > 0.0) || !(o > 0.0) || rhat == o) { phase = X_FAIL; cause = 1; return; }
        dir = rhat > o ? +1 : -1;
        // the cell (c1, c2] the scan from o ends in: c1 the last grid point before the root, c2 the first one beyond it
        c1 = o; msteps = 0;
        for (;;) {
            c2 = dir > 0 ? c1 + dcs : c1 - dcs;
            if (dir > 0 ? c2 >= rhat : c2 <= rhat) break;
            c1 = c2;
            if (++msteps > 4000) { phase = X_FAIL; cause = 2; return; }
        }
        // getsol's clamp at the floor of the scan (:463-467: the start value of the model) and its abort below it are the
        // full search's business
        if ((k > 0 && dir > 0 && o + dcs <= cc) || (dir < 0 && c2 <= cc)) { phase = X_FAIL; cause = 3; return; }
        shifted = 0;
        creq = c1; phase = X_E1;
    }

    RFS_HD void enter_nevill(double del2) {
        nv.enter(c1, del1, c2, del2, betmx);
        creq = nv.creq; phase = X_NEV;
    }

    RFS_HD void step_nevill(double f) {
        nv.advance(f);
        if (nv.phase == CellNevillT<LAZY>::N_FAIL) { phase = X_FAIL; cause = 5; return; }   // the reference rejects the root (above the fastest layer)
        if (nv.phase == CellNevillT<LAZY>::N_DONE) {
            // a period whose result is wanted must start from an origin that is the reference's to 1e-7 or better -- in effect: unless the run-up did not contract at all -- (its root then is
            // the reference's float32 value but for rare cases); a run-up that did not get there -- roots found by
            // bisections alone pass the origin's error on undiminished -- is the full search's business
            if (k >= k0 && oerr > otol) { phase = X_FAIL; cause = 7; return; }
            oerr *= nv.g3;
            cprev = nv.result; nsupplied += nv.nsupplied; phase = X_DONE; return;
        }
        creq = nv.creq;
    }

    // consume f = secular(creq) of period k; afterwards either a new request is pending, or the period is finished
    // (phase X_DONE: root() / next()), or the machine has declined
    RFS_HD void advance(double f) {
        nev++;
        if (f != f) { phase = X_FAIL; cause = 6; return; }
        if (phase == X_E1) { del1 = f; creq = c2; phase = X_E2; return; }
        if (phase == X_E1B) { del1 = f; if (!diffsign(del1, del2s)) { phase = X_FAIL; cause = 4; return; } enter_nevill(del2s); return; }
        if (phase == X_E2) {
            if (diffsign(del1, f)) { enter_nevill(f); return; }
            // no sign change in the cell the approximate root lies in: that root is within its own error of a grid point
            // and the change is in the neighbouring cell -- anything else is not the situation this machine is for
            const double tol = 4.0e-7 * fabs(rhat);
            if (shifted) { phase = X_FAIL; cause = 4; return; }
            shifted = 1;
            if (fabs(rhat - c2) <= tol) {                   // one cell further: c2 becomes c1 (value known), a new c2
                c1 = c2; del1 = f; msteps++;
                c2 = dir > 0 ? c1 + dcs : c1 - dcs;
                if (dir < 0 && c2 <= cc) { phase = X_FAIL; cause = 3; return; }
                creq = c2; phase = X_E2;
            } else if (fabs(rhat - c1) <= tol && msteps > 0) {   // one cell back: c1 becomes c2 (value known), a new c1
                c2 = c1; del2s = del1; msteps--;
                c1 = grid(msteps);
                creq = c1; phase = X_E1B;
            } else { phase = X_FAIL; cause = 4; }
            return;
        }
        step_nevill(f);
    }

    // The whole machine out of / into EXACT_SPILL_ND doubles at stride cp (k_swd_exact in rounds: a lane that has used up its
    // budget of evaluations hands its group on, in the middle of a period if need be).  Integers, flags and float32 values
    // travel as doubles (exact); the Neville table goes along.  load() installs `tab` as the table's storage.
    RFS_HD void save(double* D, size_t cp) const {
        int f = 0;
        auto put = [&](double v) { D[(size_t)f * cp] = v; f++; };
        put(nv.c1); put(nv.c2); put(nv.del1); put(nv.del2); put(nv.c3); put(nv.del3v); put(nv.creq); put(nv.result);
        put((double)nv.phase); put((double)nv.nev); put((double)nv.m); put((double)nv.nctrl); put((double)nv.nsupplied);
        put(nv.ex1 ? 1.0 : 0.0); put(nv.ex2 ? 1.0 : 0.0); put(nv.lin ? 1.0 : 0.0);
        put((double)nv.betmx); put((double)nv.g1); put((double)nv.g2); put((double)nv.g3);
        for (int i = 0; i < 24; i++) put(nv.tab.base[i * nv.tab.stride]);
        put(creq); put(omega); put(o); put(rhat); put(c1); put(c2); put(del1); put(del2s); put(cprev); put(cc); put(dcs);
        put((double)k); put((double)k0); put((double)k1); put((double)phase); put((double)dir); put((double)msteps);
        put((double)shifted); put((double)nev); put((double)cause); put((double)nsupplied);
        put((double)betmx); put((double)oerr); put((double)otol);
    }
    RFS_HD void load(const double* D, size_t cp, double* tab, int tabstride) {
        int f = 0;
        auto get = [&]() { const double v = D[(size_t)f * cp]; f++; return v; };
        nv.c1 = get(); nv.c2 = get(); nv.del1 = get(); nv.del2 = get(); nv.c3 = get(); nv.del3v = get(); nv.creq = get(); nv.result = get();
        nv.phase = (int)get(); nv.nev = (int)get(); nv.m = (int)get(); nv.nctrl = (int)get(); nv.nsupplied = (int)get();
        nv.ex1 = get() != 0.0; nv.ex2 = get() != 0.0; nv.lin = get() != 0.0;
        nv.betmx = (float)get(); nv.g1 = (float)get(); nv.g2 = (float)get(); nv.g3 = (float)get();
        nv.tab.base = tab; nv.tab.stride = tabstride;
        for (int i = 0; i < 24; i++) tab[i * tabstride] = get();
        creq = get(); omega = get(); o = get(); rhat = get(); c1 = get(); c2 = get(); del1 = get(); del2s = get(); cprev = get();
        cc = get(); dcs = get();
        k = (int)get(); k0 = (int)get(); k1 = (int)get(); phase = (int)get(); dir = (int)get(); msteps = (int)get();
        shifted = (int)get(); nev = (int)get(); cause = (int)get(); nsupplied = (int)get();
        betmx = (float)get(); oerr = (float)get(); otol = (float)get();
    }

    RFS_HD double root() const { return cprev; }             // unrounded (the next period's origin); the output is (float) of it
    RFS_HD bool wanted() const { return k >= k0; }
    // after X_DONE: on to the next period of the group, or finished (returns false)
    template <class RootFn, class OmegaFn>
    RFS_HD bool next(const RootFn& approx, const OmegaFn& om) {
        k++;
        if (k >= k1) return false;
        start_period(approx, om);
        return true;
    }
};
using ExactGroup = ExactGroupT<true>;

// The secular function of wave family F at phase velocity c, layer constants through a loader (m -> SwdLayerC):
// the arithmetic of the lanes-per-item search (raw recurrence, power-of-two rescale every eighth layer, one final
// normalisation), evaluated by ONE lane.
template <class F, bool DUAL = false, class LoadL>
RFS_HD double swd_secular_family(int n, const LoadL& loadL, double omega_raw, double c, const FmVC* vc = nullptr) {
    const double omega = omega_raw < 1.0e-4 ? 1.0e-4 : omega_raw, iomega = 1.0 / omega;
    const double wvno = omega_raw / c, wvno2 = wvno * wvno, tt = -2.0 * wvno2;
    double e[F::NV];
    F::halfspace(loadL(n - 1), wvno, wvno2, omega, iomega, e);
    // two layers per trip, their constants in two sets of registers that are loaded in turn (the next layer's are on their way
    // during this layer's arithmetic): a single rotating set costs six register copies per layer
    SwdLayerC La = loadL(n - 2 >= 0 ? n - 2 : 0);
    auto layer = [&](const SwdLayerC& L, int m) {
        double ent[F::NENT];
        if (DUAL) F::entries_dual(L, wvno, wvno2, omega, iomega, ent, vc);
        else F::entries(L, wvno, wvno2, omega, iomega, ent);
        F::apply(e, ent, tt);
        if ((m & 7) == 0) swd_rescale_pow2_n<F::NV>(e);
    };
    for (int m = n - 2; m >= 0; m -= 2) {
        const SwdLayerC Lb = loadL(m > 0 ? m - 1 : 0);
        layer(La, m);
        if (m == 0) break;
        La = loadL(m > 1 ? m - 2 : 0);
        layer(Lb, m - 1);
    }
    return swd_finish_n<F::NV>(e);
}


// ---------------------------------------------------------------------------
// Eigenfunction pass.  float32 pi as in sregn96.f90:1654.
// ---------------------------------------------------------------------------
constexpr double SR_PI32 = 3.1415927410125732;

// Vertical wavenumber nu = sqrt(nu2) of a REAL nu2 (every nu of this problem: wvno^2 - (omega/v)^2): it lies on the
// real axis (evanescent, ev) or on the imaginary one (nu = i |nu|).  One rsqrt gives |nu| and 1/|nu| (inf at 0).
struct SvNu { double nu, inu, nu2; bool ev; };
RFS_HD SvNu sv_nu(double nu2) {
    SvNu r;
    r.nu2 = nu2;
    r.ev = nu2 >= 0.0;
    const double ax = fabs(nu2);
    r.inu = rsqrt_p(ax);
    r.nu = (ax > 1.0e-290) ? ax * r.inu : 0.0;
    return r;
}
RFS_HD cplx sv_cplx(const SvNu& N) { return N.ev ? C(N.nu, 0.0) : C(0.0, N.nu); }            // nu  (principal root)
RFS_HD cplx sv_cinv(const SvNu& N) { return N.ev ? C(N.inu, 0.0) : C(0.0, -N.inu); }         // 1 / nu

// what one layer's sweeps share: the scaled hyperbolic functions of varsv, and -- for the layer integrals of the
// down-sweep -- the wavenumbers themselves and exp(-d nu) (which the functions above are made of anyway)
struct SvTrig { double cosp, rsinp, sinpr, cossv, rsinsv, sinsvr, pex, svex; SvNu na, nb; cplx ea, eb; };

// sregn96.f90:831-915 varsv for an elastic layer, specialised to the only two cases that
// occur (vertical wavenumbers are square roots of REAL numbers: pure real or pure imaginary).
// e = exp(-d nu) with gfunc's cut-off (0 beyond Re >= 75, :1348-1352).
RFS_HD void sv_trig_one(double nu2, double d, double& cosx, double& rsinx, double& sinxr, double& ex, SvNu& N, cplx& e) {
    const double tiny = (double)1.0e-5f;
    N = sv_nu(nu2);
    const double arg = N.nu * d;
    if (N.ev) {                            // evanescent: nu real
        const double eh = fm_exp(-arg);
        const double fac = (arg < 30.0) ? eh * eh : 0.0;
        cosx = 0.5 * (1.0 + fac);
        const double sh = 0.5 * (1.0 - fac);
        rsinx = N.nu * sh;
        sinxr = (arg < tiny && N.nu < tiny) ? d : sh * N.inu;
        ex = arg;
        e = C((arg < 75.0) ? eh : 0.0, 0.0);
    } else {                               // propagating: nu = i*kap
        double s, c;
        fm_sincos(arg, &s, &c);
        cosx = c;                          // pfac = exp(0) = 1
        rsinx = -N.nu * s;
        sinxr = (N.nu < tiny) ? d : s * N.inu;
        ex = 0.0;
        e = C(c, -s);
    }
}
RFS_HD void sv_trig_one(double nu2, double d, double& cosx, double& rsinx, double& sinxr, double& ex) {
    SvNu N; cplx e;
    sv_trig_one(nu2, d, cosx, rsinx, sinxr, ex, N, e);
}

// ia, ib: reciprocals of the layer's velocities
RFS_HD void sv_trig(double wvno2, double omega, double ia, double ib, double d, SvTrig& t) {
    double xka = omega * ia, xkb = omega * ib;
    sv_trig_one(wvno2 - xka * xka, d, t.cosp, t.rsinp, t.sinpr, t.pex, t.na, t.ea);
    sv_trig_one(wvno2 - xkb * xkb, d, t.cossv, t.rsinsv, t.sinsvr, t.svex, t.nb, t.eb);
}

// per (period, chain) item: the scalars every layer step divides by, as reciprocals
struct SrItem {
    double omega, wvno, wvno2, om2, iwvno, iwvno2, iom2, iom;
    RFS_HD SrItem(double om, double k) : omega(om), wvno(k), wvno2(k * k), om2(om * om) {
        iwvno = rcp_p(k); iwvno2 = iwvno * iwvno; iom = rcp_p(om); iom2 = iom * iom;
    }
};

// cd <- normalise(cd . CA) with CA the hspec96-convention compound matrix (sregn96.f90:494-650);
// returns the log of the normalisation (normc :1405-1434).  Divisions: one reciprocal of rho om^2, one of the
// (float32) rho^2 times om^4, one of the norm; everything else multiplies by the item's reciprocals.
RFS_HD double sr_compound_step(double cd[5], const SvTrig& t, float rhof, float bf, const SrItem& Q) {
    const double wvno = Q.wvno, wvno2 = Q.wvno2, om2 = Q.om2;
    double exa = t.pex + t.svex;
    double a0 = (exa < 60.0) ? fm_exp(-exa) : 0.0;
    double cpcq = t.cosp * t.cossv, cpy = t.cosp * t.sinsvr, cpz = t.cosp * t.rsinsv;
    double cqw = t.cossv * t.sinpr, cqx = t.cossv * t.rsinp;
    double xy = t.rsinp * t.sinsvr, xz = t.rsinp * t.rsinsv, wy = t.sinpr * t.sinsvr, wz = t.sinpr * t.rsinsv;
    double rho = (double)rhof, rho2 = (double)(rhof * rhof);      // float32 rho*rho (:508,596)
    double gam = (double)((2.0f * bf) * bf) * wvno2 * Q.iom2;     // float32 2*b*b (:597)
    double gam2 = gam * gam, gamm1 = gam - 1.0, gamm2 = gamm1 * gamm1;
    double cqww2 = cqw * wvno2, cqxw2 = cqx * Q.iwvno2, gg1 = gam * gamm1;
    double a0c = 2.0 * (a0 - cpcq);
    double xz2 = xz * Q.iwvno2, gxz2 = gam * xz2, g2xz2 = gam2 * xz2;
    double a0cgg1 = a0c * (gam + gamm1);
    double wy2 = wy * wvno2, g2wy2 = gamm2 * wy2, g1wy2 = gamm1 * wy2;
    double rom = rho * om2, irom = rcp_p(rom), r2o4 = rho2 * om2 * om2, ir2o4 = rcp_p(r2o4);
    double temp = a0c * gg1 + g2xz2 + g2wy2;
    double c33 = a0 + temp + temp;
    double c11 = cpcq - temp;
    double c12 = (-cqx + wvno2 * cpy) * irom;
    temp = 0.5 * a0cgg1 + gxz2 + g1wy2;
    double c13 = wvno * temp * irom;
    double c14 = (-cqww2 + cpz) * irom;
    temp = wvno2 * (a0c + wy2) + xz;
    double c15 = -temp * ir2o4;
    double c21 = (-gamm2 * cqw + gam2 * cpz * Q.iwvno2) * rom;
    double c22 = cpcq;
    double c23 = (gamm1 * cqww2 - gam * cpz) * Q.iwvno;
    double c24 = -wz;
    temp = 0.5 * a0cgg1 * gg1 + gam2 * gxz2 + gamm2 * g1wy2;
    double c31 = -2.0 * temp * rom * Q.iwvno;
    double c32 = -wvno * (gam * cqxw2 - gamm1 * cpy) * 2.0;
    double c34 = -2.0 * c23, c35 = -2.0 * c13;
    double c41 = (-gam2 * cqxw2 + gamm2 * cpy) * rom;
    double c42 = -xy;
    double c43 = -0.5 * c32;
    temp = gamm2 * (a0c * gam2 + g2wy2) + gam2 * g2xz2;
    double c51 = -r2o4 * temp * Q.iwvno2;
    double c53 = -0.5 * c31;
    // ca(2,5)=ca(1,4) ca(4,4)=ca(2,2) ca(4,5)=ca(1,2) ca(5,2)=ca(4,1) ca(5,4)=ca(2,1) ca(5,5)=ca(1,1)
    double n0 = cd[0] * c11 + cd[1] * c21 + cd[2] * c31 + cd[3] * c41 + cd[4] * c51;
    double n1 = cd[0] * c12 + cd[1] * c22 + cd[2] * c32 + cd[3] * c42 + cd[4] * c41;
    double n2 = cd[0] * c13 + cd[1] * c23 + cd[2] * c33 + cd[3] * c43 + cd[4] * c53;
    double n3 = cd[0] * c14 + cd[1] * c24 + cd[2] * c34 + cd[3] * c22 + cd[4] * c21;
    double n4 = cd[0] * c15 + cd[1] * c14 + cd[2] * c35 + cd[3] * c12 + cd[4] * c11;
    double t1 = fmax(fmax(fmax(fabs(n0), fabs(n1)), fmax(fabs(n2), fabs(n3))), fabs(n4));
    if (t1 < 1.0e-40) t1 = 1.0;
    const double it1 = rcp_p(t1);
    cd[0] = n0 * it1; cd[1] = n1 * it1; cd[2] = n2 * it1; cd[3] = n3 * it1; cd[4] = n4 * it1;
    return log(t1);
}

// vv <- normalise(AA . vv) with AA the Haskell matrix (sregn96.f90:917-991); returns log-norm.
RFS_HD double sr_haskell_step(double vv[4], const SvTrig& t, float rhof, float bf, const SrItem& Q) {
    const double wvno = Q.wvno, wvno2 = Q.wvno2, om2 = Q.om2;
    double dfac = ((t.pex - t.svex) > 70.0) ? 0.0 : fm_exp(t.svex - t.pex);
    double cossv = dfac * t.cossv, rsinsv = dfac * t.rsinsv, sinsvr = dfac * t.sinsvr;
    double cosp = t.cosp, rsinp = t.rsinp, sinpr = t.sinpr;
    double gam = (double)((2.0f * bf) * bf) * wvno2 * Q.iom2, gamm1 = gam - 1.0;
    double rom = (double)rhof * om2, irom = rcp_p(rom);
    double a11 = cossv + gam * (cosp - cossv);
    double a12 = -wvno * gamm1 * sinpr + gam * rsinsv * Q.iwvno;
    double a13 = -wvno * (cosp - cossv) * irom;
    double a14 = (wvno2 * sinpr - rsinsv) * irom;
    double a21 = gam * rsinp * Q.iwvno - wvno * gamm1 * sinsvr;
    double a22 = cosp - gam * (cosp - cossv);
    double a23 = (-rsinp + wvno2 * sinsvr) * irom;
    double a31 = rom * gam * gamm1 * (cosp - cossv) * Q.iwvno;
    double a32 = rom * (-gamm1 * gamm1 * sinpr + gam * gam * rsinsv * Q.iwvno2);
    double a41 = rom * (gam * gam * rsinp * Q.iwvno2 - gamm1 * gamm1 * sinsvr);
    // a24=-a13 a33=a22 a34=-a12 a42=-a31 a43=-a21 a44=a11
    double n0 = a11 * vv[0] + a12 * vv[1] + a13 * vv[2] + a14 * vv[3];
    double n1 = a21 * vv[0] + a22 * vv[1] + a23 * vv[2] - a13 * vv[3];
    double n2 = a31 * vv[0] + a32 * vv[1] + a22 * vv[2] - a12 * vv[3];
    double n3 = a41 * vv[0] - a31 * vv[1] - a21 * vv[2] + a11 * vv[3];
    double t1 = fmax(fmax(fabs(n0), fabs(n1)), fmax(fabs(n2), fabs(n3)));
    if (t1 < 1.0e-40) t1 = 1.0;
    const double it1 = rcp_p(t1);
    vv[0] = n0 * it1; vv[1] = n1 * it1; vv[2] = n2 * it1; vv[3] = n3 * it1;
    return log(t1);
}

// The same two steps through a FLUID layer (dnka :555-575, hska :931-945; varsv's fluid branch :858-877 is the P
// half of the elastic one): cosp, rsinp, sinpr, pex from sv_trig_one.
RFS_HD double sr_compound_step_fluid(double cd[5], double cosp, double rsinp, double sinpr, double pex, float rhof, double om2) {
    const double dfac = (pex > 35.0) ? 0.0 : fm_exp(-pex);
    const double rom = (double)rhof * om2;
    const double c12 = -rsinp / rom, c21 = -(double)rhof * sinpr * om2;
    double n0 = cd[0] * cosp + cd[1] * c21;
    double n1 = cd[0] * c12 + cd[1] * cosp;
    double n2 = cd[2] * dfac;
    double n3 = cd[3] * cosp + cd[4] * c21;
    double n4 = cd[3] * c12 + cd[4] * cosp;
    double t1 = fmax(fmax(fmax(fabs(n0), fabs(n1)), fmax(fabs(n2), fabs(n3))), fabs(n4));
    if (t1 < 1.0e-40) t1 = 1.0;
    cd[0] = n0 / t1; cd[1] = n1 / t1; cd[2] = n2 / t1; cd[3] = n3 / t1; cd[4] = n4 / t1;
    return log(t1);
}
RFS_HD double sr_haskell_step_fluid(double vv[4], double cosp, double rsinp, double sinpr, double pex, float rhof, double om2) {
    const double dfac = (pex > 35.0) ? 0.0 : fm_exp(-pex);
    const double rom = (double)rhof * om2;
    const double a23 = -rsinp / rom, a32 = -rom * sinpr;
    double n0 = dfac * vv[0];
    double n1 = cosp * vv[1] + a23 * vv[2];
    double n2 = a32 * vv[1] + cosp * vv[2];
    double n3 = dfac * vv[3];
    double t1 = fmax(fmax(fabs(n0), fabs(n1)), fmax(fabs(n2), fabs(n3)));
    if (t1 < 1.0e-40) t1 = 1.0;
    vv[0] = n0 / t1; vv[1] = n1 / t1; vv[2] = n2 / t1; vv[3] = n3 / t1;
    return log(t1);
}

// Half-space compound vector (evalg, jbdry=0, elastic: sregn96.f90:760-776), real parts (up :442-446).
RFS_HD void sr_halfspace_vector(double a, double b, double rho, double wvno, double om, double cd[5]) {
    double wvno2 = wvno * wvno, om2 = om * om;
    double xka = om / a, xkb = om / b;
    cplx ra = csqrt_p(C(wvno2 - xka * xka)), rb = csqrt_p(C(wvno2 - xkb * xkb));
    double gam = b * wvno / om; gam = 2.0 * (gam * gam);
    double gamm1 = gam - 1.0;
    cplx rarb = ra * rb;
    cplx den = (-rho * rho * om2 * om2 * wvno2) * rarb;
    cplx g0 = (rho * rho) * om2 * om2 * ((-gam * gam) * rarb + wvno2 * gamm1 * gamm1);
    cplx g1 = (-rho * wvno2 * om2) * ra;
    cplx g2 = (-rho) * ((-gam) * rarb + wvno2 * gamm1) * (om2 * wvno);
    cplx g3 = (rho * wvno2 * om2) * rb;
    cplx g4 = wvno2 * (wvno2 - rarb);
    cplx iden = inv(den);
    cd[0] = (0.25 * g0 * iden).re; cd[1] = (0.25 * g1 * iden).re; cd[2] = (0.25 * g2 * iden).re;
    cd[3] = (0.25 * g3 * iden).re; cd[4] = (0.25 * g4 * iden).re;
}

// Up-sweep (sregn96.f90:404-492): Store(m, cd[5], exe) is called for m = n-1 .. 0.
// WATER: the top layer may be a fluid (vs = 0: the one place surfdisp96 can find roots for, :138-139, :870-886).
template <bool WATER = false, class Mdl, class StoreFn>
RFS_HD void sr_up(const Mdl& M, double omega, double wvno, const StoreFn& store) {
    const int n = M.n;
    const SrItem Q(omega, wvno);
    double cd[5];
    sr_halfspace_vector(M.A(n - 1), M.B(n - 1), M.R(n - 1), wvno, omega, cd);
    double exsum = 0.0;
    store(n - 1, cd, 0.0);
    for (int m = n - 2; m >= 0; m--) {
        if (WATER && m == 0 && M.B(0) <= 0.0) {
            const double xka = omega / M.A(0);
            double cosp, rsinp, sinpr, pex;
            sv_trig_one(Q.wvno2 - xka * xka, M.D(0), cosp, rsinp, sinpr, pex);
            double exn = sr_compound_step_fluid(cd, cosp, rsinp, sinpr, pex, M.Rf(0), Q.om2);
            exsum = exsum + pex + exn;
            store(m, cd, exsum);
            continue;
        }
        SvTrig t;
        sv_trig(Q.wvno2, omega, rcp_p(M.A(m)), rcp_p(M.B(m)), M.D(m), t);
        double exn = sr_compound_step(cd, t, M.Rf(m), M.Bf(m), Q);
        exsum = exsum + t.pex + t.svex + exn;
        store(m, cd, exsum);
    }
}

struct Eig4 { double ur, uz, tz, tr; };

// E and E^-1 of an elastic layer plus the closed-form integrals shared by the six
// intijr calls (sregn96.f90:719-758, 1203-1323, 1325-1403).  na, nb: the layer's vertical wavenumbers; ea, eb:
// exp(-d nu) of both (inner layers only), as sv_trig leaves them.
struct LayerInt { double i11, i13, i22, i24, i33, i44; };

RFS_HD void sr_layer_integrals(double b, double rho, double irho, double d, bool halfspace, const SrItem& Q,
                               const SvNu& na, const SvNu& nb, cplx ea, cplx eb,
                               const Eig4& top, const Eig4& bot, LayerInt& I) {
    const double wvno = Q.wvno;
    const cplx ra = sv_cplx(na), rb = sv_cplx(nb), ira = sv_cinv(na), irb = sv_cinv(nb);
    double gam = b * wvno * Q.iom; gam = 2.0 * (gam * gam);
    double gamm1 = gam - 1.0;
    double rom = rho * Q.om2, irom = irho * Q.iom2;
    // E (evalg :719-737), columns [PU, SvU, PD, SvD], rows Ur, Uz, Tz, Tr:
    //   Ur: ( k,      nu_b,       k,    -nu_b     )      Uz: ( nu_a,      k,    -nu_a,       k   )
    //   Tz: ( e31,  rgk nu_b,   e31,  -rgk nu_b   )      Tr: ( rgk nu_a, e31,  -rgk nu_a,   e31  )
    const double e31 = rom * gamm1, rgk = rom * gam * Q.iwvno;
    // E^-1 rows 1..4 (columns Ur, Uz, Tz, Tr)
    double hg = 0.5 * gam * Q.iwvno, hr = 0.5 * irom;
    const double hk = 0.5 * wvno * irom;
    cplx i12 = (-0.5 * gamm1) * ira, i14 = hk * ira;
    cplx i21 = (-0.5 * gamm1) * irb, i23 = hk * irb;
    // downgoing potentials at the top of the layer (rows 3,4)
    const cplx D = C(hg * top.ur) - i12 * top.uz - C(hr * top.tz) - i14 * top.tr;      // km1pd
    const cplx W = C(hg * top.uz) - i21 * top.ur - i23 * top.tz - C(hr * top.tr);      // km1sd
    // intijr's integrand for rows i, j of E is quadratic in the four potentials, with E(i,3) = s_i E(i,1) and
    // E(i,4) = t_i E(i,2), (s, t) = (+,-) for Ur, Tz and (-,+) for Uz, Tr.  The six integrals the energy sums need pair rows
    // of the SAME sign class, so they share three numbers per class -- the P-P, S-S and P-S weights below -- and differ only
    // in REAL coefficients (nu_a^2, nu_b^2 are real): 18 complex products per layer instead of 6 x 22.
    double reAp, reAm, reBp, reBm, rC, rCp;
    if (!halfspace) {
        const cplx U = C(hg * bot.ur) + i12 * bot.uz - C(hr * bot.tz) + i14 * bot.tr;  // kmpu
        const cplx V = i21 * bot.ur + C(hg * bot.uz) + i23 * bot.tz - C(hr * bot.tr);  // kmsu
        // f, g, h1, h2 with the reference's guards (ea, eb arrive with gfunc's cut-off at 75 applied)
        cplx FA, GA, FB, GB, H1, H2;
        cplx ea40 = (ra.re * d < 40.0) ? ea : C(0.0), eb40 = (rb.re * d < 40.0) ? eb : C(0.0);
        FA = (na.nu < 1.0e-8) ? C(d) : (1.0 - ea40 * ea40) * (0.5 * ira);
        FB = (nb.nu < 1.0e-8) ? C(d) : (1.0 - eb40 * eb40) * (0.5 * irb);
        GA = d * ea; GB = d * eb;
        cplx rsum = ra + rb, rdif = ra - rb;
        cplx esum = ((rsum.re * d) < 40.0) ? ea * eb : C(0.0);
        H1 = (sqrt(norm2(rsum)) < 1.0e-8) ? C(d) : (1.0 - esum) * inv(rsum);
        H2 = (sqrt(norm2(rdif)) < 1.0e-8) ? C(d) : (eb40 - ea40) * inv(rdif);
        const double aF = re_mul(U * U + D * D, FA), aG = 2.0 * re_mul(U * D, GA);
        const double bF = re_mul(V * V + W * W, FB), bG = 2.0 * re_mul(V * W, GB);
        reAp = aF + aG; reAm = aF - aG;
        reBp = bF + bG; reBm = bF - bG;
        const cplx h1 = (U * V - D * W) * H1, uw = U * W, dv = D * V;
        rC = re_mul(rb, h1 + (dv - uw) * H2);
        rCp = re_mul(ra, h1 + (uw - dv) * H2);
    } else {
        const cplx fa = 0.5 * ira, fb = 0.5 * irb, fab = inv(ra + rb);
        reAp = reAm = re_mul(D * D, fa);
        reBp = reBm = re_mul(W * W, fb);
        const cplx cm = -((D * W) * fab);
        rC = re_mul(rb, cm); rCp = re_mul(ra, cm);
    }
    const double na2 = na.nu2, nb2 = nb.nu2, e3 = e31;                 // e31 = e42 = rho om^2 (gamma - 1)
    I.i11 = wvno * wvno * reAp + nb2 * reBm + 2.0 * wvno * rC;
    I.i33 = e3 * e3 * reAp + (rgk * rgk) * nb2 * reBm + 2.0 * (e3 * rgk) * rC;
    I.i13 = wvno * e3 * reAp + rgk * nb2 * reBm + (wvno * rgk + e3) * rC;
    I.i22 = na2 * reAm + wvno * wvno * reBp + 2.0 * wvno * rCp;
    I.i44 = (rgk * rgk) * na2 * reAm + e3 * e3 * reBp + 2.0 * (rgk * e3) * rCp;
    I.i24 = rgk * na2 * reAm + wvno * e3 * reBp + (e3 + wvno * rgk) * rCp;
}

// Interface term of getdcdh (sregn96.f90:1436-1535) WITHOUT the final `fac` (m == 0: "above" is vacuum).
// il2m_*, imu_*: reciprocals of lambda + 2 mu and of mu on the two sides (imu only used where mu != 0).
RFS_HD double sr_interface_term(bool top_surface, double rho_m, double mu_m, double lam_m, double il2m_m, double imu_m,
                                double rho_u, double mu_u, double lam_u, double il2m_u, double imu_u, const Eig4& u,
                                double om2, double wvno, double wvno2, bool fluid_above = false) {
    double tur = u.ur, tuz = u.uz, ttz = u.tz, ttr = u.tr;
    if (fluid_above) {                       // solid under the water layer: Ur jumps across the interface, :1504-1508
        const double xl2mp = lam_m + mu_m + mu_m, xl2mm = lam_u + mu_u + mu_u;
        const double duzdzp = (ttz + wvno * lam_m * tur) / xl2mp;
        const double durdzp = (ttr / mu_m) - wvno * tuz;
        const double urb = -wvno * ttz / (rho_u * om2);
        const double drur2 = tur * tur * rho_m - urb * urb * rho_u;
        const double dlur2 = tur * tur * xl2mp - urb * urb * xl2mm;
        const double duzdzm = (ttz + wvno * lam_u * urb) / lam_u;
        const double durdzm = wvno * tuz;
        const double g1 = om2 * (rho_m - rho_u) * tuz * tuz, g2 = om2 * drur2;
        const double g3 = -wvno2 * (mu_m - mu_u) * tuz * tuz, g4 = -wvno2 * dlur2;
        const double g5 = xl2mp * duzdzp * duzdzp - xl2mm * duzdzm * duzdzm;
        const double g6 = mu_m * durdzp * durdzp - mu_u * durdzm * durdzm;
        return g1 + g2 + g3 + g4 + g5 + g6;
    }
    double xl2mp = lam_m + mu_m + mu_m;
    double duzdzp = (ttz + wvno * lam_m * tur) * il2m_m;
    double durdzp = (mu_m == 0.0) ? wvno * tuz : (ttr * imu_m) - wvno * tuz;
    double drho, dmu, dl2mu, g5, g6;
    if (top_surface) {
        drho = rho_m; dmu = mu_m; dl2mu = lam_m + mu_m + mu_m;
        g5 = xl2mp * duzdzp * duzdzp;
        g6 = mu_m * durdzp * durdzp;
    } else {
        drho = rho_m - rho_u; dmu = mu_m - mu_u;
        dl2mu = (lam_m - lam_u) + dmu + dmu;
        double xl2mm = lam_u + mu_u + mu_u;
        double durdzm = (mu_u == 0.0) ? wvno * tuz : (ttr * imu_u) - wvno * tuz;
        double duzdzm = (ttz + wvno * lam_u * tur) * il2m_u;
        g5 = xl2mp * duzdzp * duzdzp - xl2mm * duzdzm * duzdzm;
        g6 = mu_m * durdzp * durdzp - mu_u * durdzm * durdzm;
    }
    double g1 = om2 * drho * tuz * tuz;
    double g2 = om2 * (tur * tur * drho);
    double g3 = -wvno2 * dmu * tuz * tuz;
    double g4 = -wvno2 * (tur * tur * dl2mu);
    return g1 + g2 + g3 + g4 + g5 + g6;
}

// Down-sweep fused with eigenfunctions, energy integrals and interface terms.
// Load(m, cd[5], exe) returns what sr_up stored.  Emit(m, da, db, dr, dh) receives the RAW
// per-layer factors (before the 1/(U I0) and `fac` scalings); the caller rescales them with
// the returned (ugr, sumi0, fac) -- see sr_finish_scale.
struct SrTotals { double ugr, sumi0, fac; };

template <bool WATER = false, class Mdl, class LoadFn, class EmitFn>
RFS_HD SrTotals sr_down_energy(const Mdl& M, double omega, double wvno, const LoadFn& load,
                               const EmitFn& emit) {
    const int n = M.n;
    const SrItem Q(omega, wvno);
    const double om2 = Q.om2, wvno2 = Q.wvno2, c = omega * Q.iwvno;
    double cd[5], exe0, exe_m;
    load(0, cd, exe0);
    const double f1213 = -cd[1], if1213 = rcp_p(f1213);
    Eig4 top{cd[2] / cd[1], 1.0, 0.0, 0.0};
    const bool wat0 = WATER && M.B(0) <= 0.0;
    if (wat0) top.ur = 0.0;                  // svfunc :319-330 (Ur, Tr of the fluid's top), energy :1140 (Ur from Tz = 0)
    double vv[4] = {1.0, 0.0, 0.0, 0.0};
    double exa = 0.0;
    double sumi0 = 0.0, sumi1 = 0.0, sumi2 = 0.0, sumi3 = 0.0;
    double rho_u = 0.0, mu_u = 0.0, lam_u = 0.0, il2m_u = 0.0, imu_u = 0.0;
    for (int m = 0; m < n; m++) {
        const bool half = (m == n - 1);
        double a = M.A(m), b = M.B(m), rho = M.R(m), d = M.D(m);
        double mu = rho * (b * b), lam = rho * (a * a) - 2 * mu;
        Eig4 bot = top;
        const bool fluid = wat0 && m == 0;
        SvTrig t;
        const double ia = rcp_p(a), ib = fluid ? 0.0 : rcp_p(b), irho = rcp_p(rho);
        if (!half) load(m + 1, cd, exe_m);       // (issued here: the Haskell step below hides the scratch round trip)
        if (!half) {
            if (fluid) {
                const double xka = omega / a;
                double cosp, rsinp, sinpr, pex;
                sv_trig_one(wvno2 - xka * xka, d, cosp, rsinp, sinpr, pex);
                double ex2 = sr_haskell_step_fluid(vv, cosp, rsinp, sinpr, pex, M.Rf(m), om2);
                exa = exa + pex + ex2;
            } else {
                sv_trig(wvno2, omega, ia, ib, d, t);
                double ex2 = sr_haskell_step(vv, t, M.Rf(m), M.Bf(m), Q);
                exa = exa + t.pex + ex2;
            }
            // svfunc :283-315
            double cd1 = cd[0], cd2 = cd[1], cd3 = cd[2], cd4 = -cd[2], cd5 = cd[3], cd6 = cd[4];
            double tz1 = -vv[3], tz2 = -vv[2], tz3 = vv[1], tz4 = vv[0];
            double uu1 = tz2 * cd6 - tz3 * cd5 + tz4 * cd4;
            double uu2 = -tz1 * cd6 + tz3 * cd3 - tz4 * cd2;
            double uu3 = tz1 * cd5 - tz2 * cd3 + tz4 * cd1;
            double uu4 = -tz1 * cd4 + tz2 * cd2 - tz3 * cd1;
            double ext = exa + exe_m - exe0;
            if (ext > -80.0 && ext < 80.0) {
                double fact = fm_exp(ext) * if1213;
                bot = Eig4{uu1 * fact, uu2 * fact, uu3 * fact, uu4 * fact};
            } else {
                bot = Eig4{0.0, 0.0, 0.0, 0.0};
            }
        } else {
            const double xka = omega * ia, xkb = omega * ib;
            t.na = sv_nu(wvno2 - xka * xka); t.nb = sv_nu(wvno2 - xkb * xkb);
            t.ea = C(0.0); t.eb = C(0.0);
        }
        if (fluid) {
            // fluid layer: 2 x 2 potentials (evalg :800-811, intijr :1245-1262), getmat :1551-1565, energy :1122-1140
            const double xka = omega / a, rom = rho * om2;
            const cplx ra = csqrt_p(C(wvno2 - xka * xka)), ira = inv(ra);
            const cplx km1pd = (-0.5 * top.uz) * ira - C(0.5 / rom * top.tz);
            const cplx kmpu = (0.5 * bot.uz) * ira - C(0.5 / rom * bot.tz);
            const cplx ea = (ra.re * d < 75.0) ? cexp_p(-(d * ra)) : C(0.0);
            const cplx ea40 = (ra.re * d < 40.0) ? ea : C(0.0);
            const cplx FA = (sqrt(norm2(ra)) < 1.0e-8) ? C(d) : (1.0 - ea40 * ea40) * (0.5 * ira);
            const cplx GA = d * ea;
            const cplx uu = kmpu * kmpu * FA, ud = kmpu * km1pd * GA, dd = km1pd * km1pd * FA;
            const double i11 = (ra * ra * (uu - 2.0 * ud + dd)).re;          // e(1,1) = ra, e(1,2) = -ra
            const double i22 = (rom * rom) * (uu + 2.0 * ud + dd).re;        // e(2,1) = e(2,2) = -rho om2
            const double TA = rho * a * a, a12 = -(wvno2 - om2 / (a * a)) / rom, kr = wvno / rom;
            const double URUR = i22 * kr * kr, UZUZ = i11, URDUZ = -kr * a12 * i22, DUZDUZ = a12 * a12 * i22;
            sumi0 += rho * (URUR + UZUZ);
            sumi1 += TA * URUR;
            sumi2 -= TA * URDUZ;                                             // TF = TA (TN = 0)
            sumi3 += TA * DUZDUZ;
            const double facah = rho * a * (URUR - 2. * URDUZ / wvno), facav = rho * a * DUZDUZ / wvno2;
            const double facr = -0.5 * c * c * (URUR + UZUZ);
            const double il2m = 1.0 / (lam + mu + mu);
            const double dh = sr_interface_term(true, rho, mu, lam, il2m, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, top, om2, wvno, wvno2);
            emit(m, facah + facav, 0.0, 0.5 * (a * facav + a * facah) / rho + facr, dh);     // (dcdb: never assigned there)
            rho_u = rho; mu_u = mu; lam_u = lam; il2m_u = il2m; imu_u = 0.0;
            top = bot;
            continue;
        }
        LayerInt I;
        sr_layer_integrals(b, rho, irho, d, half, Q, t.na, t.nb, t.ea, t.eb, top, bot, I);
        // getmat :1566-1586 (isotropic)
        double TL = rho * b * b, TC = rho * a * a, TA = TC, TF = TA - 2. * TL;
        const double iTL = irho * (ib * ib), iTC = irho * (ia * ia);
        double a12 = -wvno, a14 = iTL, a21 = wvno * TF * iTC, a23 = iTC;
        double URUR = I.i11, UZUZ = I.i22;
        double DURDUR = a12 * a12 * I.i22 + 2. * a12 * a14 * I.i24 + a14 * a14 * I.i44;
        double DUZDUZ = a21 * a21 * I.i11 + 2. * a21 * a23 * I.i13 + a23 * a23 * I.i33;
        double URDUZ = a21 * I.i11 + a23 * I.i13;
        double UZDUR = a12 * I.i22 + a14 * I.i24;
        sumi0 += rho * (URUR + UZUZ);
        sumi1 += TL * UZUZ + TA * URUR;
        sumi2 += TL * UZDUR - TF * URDUZ;
        sumi3 += TL * DURDUR + TC * DUZDUZ;
        double facah = rho * a * (URUR - 2. * URDUZ * Q.iwvno);
        double facav = rho * a * DUZDUZ * Q.iwvno2;
        double facbv = rho * b * (UZUZ + 2. * UZDUR * Q.iwvno + DURDUR * Q.iwvno2 + 4. * URDUZ * Q.iwvno);
        double facr = -0.5 * c * c * (URUR + UZUZ);
        double da = facah + facav, db = facbv;
        double dr = 0.5 * (a * facav + a * facah + b * facbv) * irho + facr;
        double dh = sr_interface_term(m == 0, rho, mu, lam, iTC, iTL, rho_u, mu_u, lam_u, il2m_u, imu_u, top, om2, wvno, wvno2,
                                      wat0 && m == 1);
        emit(m, da, db, dr, dh);
        rho_u = rho; mu_u = mu; lam_u = lam; il2m_u = iTC; imu_u = iTL;
        top = bot;
    }
    SrTotals T;
    T.ugr = (wvno * sumi1 + sumi2) / (omega * sumi0);
    double are = wvno / (2.0 * omega * T.ugr * sumi0);
    T.sumi0 = sumi0;
    T.fac = are * c / wvno2;
    return T;
}

// ---------------------------------------------------------------------------
// Love eigenfunctions and kernels (slegn96.f90): varl :248-328, up :372-445, shfunc :179-246,
// energy :447-629 -- all-solid model.  Same two-sweep layout as the Rayleigh pass: the up-sweep
// stores (uu, tt, exl) per layer, the top-down sweep rebuilds the true amplitudes, the layer
// integrals and the interface terms with O(1) state.
// ---------------------------------------------------------------------------
struct SlVar { double cosq, yl, zl, mu, rb, xkb, eexl; };

template <class Mdl>
RFS_HD void sl_varl(const Mdl& M, int m, double omega, double wvno, double dpth, SlVar& v) {
    double b = M.B(m);
    v.xkb = omega / b;
    v.rb = sqrt((wvno + v.xkb) * fabs(wvno - v.xkb));
    double q = v.rb * dpth;
    v.mu = M.R(m) * b * b;
    v.eexl = 0.0;
    if (wvno < v.xkb) {
        double sn, cs; sincos(q, &sn, &cs);
        v.yl = sn / v.rb; v.zl = -v.rb * sn; v.cosq = cs;
    } else if (wvno == v.xkb) {
        v.cosq = 1.0; v.yl = dpth; v.zl = 0.0;
    } else {
        v.eexl = q;
        double fac = (q < 18.0) ? exp(-2.0 * q) : 0.0;
        v.cosq = (1.0 + fac) * 0.5;
        double sinq = (1.0 - fac) * 0.5;
        v.yl = sinq / v.rb; v.zl = v.rb * sinq;
    }
}

// Store(m, uu, tt, exl) for m = n-1 .. 0
template <class Mdl, class StoreFn>
RFS_HD void sl_up(const Mdl& M, double omega, double wvno, const StoreFn& store) {
    const int n = M.n;
    SlVar v;
    double uu = 1.0, tt = 0.0;
    if (M.B(n - 1) > 0.01) {
        sl_varl(M, n - 1, omega, wvno, 0.0, v);
        double bl = M.B(n - 1);
        tt = -(M.R(n - 1) * (bl * bl)) * v.rb;                 // -xmu(mmax)*rb
    }
    store(n - 1, uu, tt, 0.0);
    for (int k = n - 2; k >= 0; k--) {
        sl_varl(M, k, omega, wvno, M.D(k), v);
        double a11 = v.cosq, a12 = -(v.yl / v.mu), a21 = -(v.zl * v.mu);
        double amp0 = a11 * uu + a12 * tt;
        double str0 = a21 * uu + a11 * tt;
        double rr = fmax(fabs(amp0), fabs(str0));
        if (rr < 1.0e-30) rr = 1.0;
        uu = amp0 / rr; tt = str0 / rr;
        store(k, uu, tt, log(rr) + v.eexl);
    }
}

struct SlTotals { double ugr, sumi1, fac; };

// Load(m, uu, tt, exl) returns what sl_up stored.  Emit(m, db, dr, dh) receives dc/db and dc/drho BEFORE the
// division by I1 and the interface term before `fac` (slegn96.f90:564-607); the caller rescales.
template <class Mdl, class LoadFn, class EmitFn>
RFS_HD SlTotals sl_down_energy(const Mdl& M, double omega, double wvno, const LoadFn& load, const EmitFn& emit) {
    const int n = M.n;
    const double c = omega / wvno, omega2 = omega * omega, wvno2 = wvno * wvno;
    // shfunc :212-240: amplitudes from the extended floating-point form, normalised to uu(1) (or, if that
    // vanishes, to the largest amplitude)
    double u0, t0, e0;
    load(0, u0, t0, e0);
    double umax = u0;
    if (u0 == 0.0) {
        double ext = 0.0, eprev = e0;
        for (int k = 1; k < n; k++) {
            double uk, tk, ek; load(k, uk, tk, ek);
            ext += eprev; eprev = ek;
            double fact = (ext < 80.0) ? 1. / exp(ext) : 0.0;
            uk *= fact;
            if (fabs(uk) > fabs(umax)) umax = uk;
        }
    }
    const bool scale = fabs(umax) > 0.0;
    double ut = scale ? u0 / umax : u0, tt_top = 0.0;          // tt(1) = 0 is forced (:214)
    double ext = 0.0, eprev = e0;
    double sumi0 = 0.0, sumi1 = 0.0, sumi2 = 0.0;
    double rho_u = 0.0, mu_u = 0.0;
    for (int k = 0; k < n; k++) {
        double b = M.B(k), drho = M.R(k), dpth = M.D(k);
        double TN = drho * b * b, dmu = drho * (b * b);         // TN = zrho*zb*zb, xmu = zrho*zb**2
        double ub = 0.0, tb = 0.0;
        if (k < n - 1) {
            double ek; load(k + 1, ub, tb, ek);
            ext += eprev; eprev = ek;
            double fact = (ext < 80.0) ? 1. / exp(ext) : 0.0;
            ub *= fact; tb *= fact;
            if (scale) { ub /= umax; tb /= umax; }
        }
        SlVar v;
        sl_varl(M, k, omega, wvno, dpth, v);
        double rb = v.rb < 1.0e-10 ? 1.0e-10 : v.rb;
        double upup, dupdup;
        if (k == n - 1) {
            upup = (0.5 / rb) * ut * ut;
            dupdup = (0.5 * rb) * ut * ut;
        } else {
            cplx nub = (wvno < v.xkb) ? cplx{0.0, rb} : cplx{rb, 0.0};
            cplx xnub = dmu * nub;
            cplx iw = inv(wvno * xnub);
            double h = 0.5 / wvno;
            cplx km1dn = C(h * ut) - (0.5 * iw) * tt_top;       // einvl(2,1) uu(k) + einvl(2,2) tt(k)
            cplx kmup = C(h * ub) + (0.5 * iw) * tb;            // einvl(1,1) uu(k1) + einvl(1,2) tt(k1)
            cplx f3 = dpth * nub;
            cplx ex2 = (f3.re < 40.0) ? cexp_p(-2.0 * f3) : C(0.0);
            cplx f = (1.0 - ex2) * inv(2.0 * nub);
            cplx ex1 = (f3.re < 75.0) ? cexp_p(-f3) : C(0.0);
            cplx g = dpth * ex1;
            cplx f1 = f * ((wvno * wvno) * (kmup * kmup) + (wvno * wvno) * (km1dn * km1dn));
            cplx f2 = g * (2.0 * (wvno * wvno)) * (kmup * km1dn);
            upup = (f1 + f2).re;
            dupdup = (nub * nub * (f1 - f2)).re;
        }
        sumi0 += drho * upup; sumi1 += TN * upup; sumi2 += TN * dupdup;
        double db = c * drho * b * upup + c * drho * b * dupdup / wvno2;
        double dr = 0.5 * c * (-c * c * upup + b * b * upup + b * b * dupdup / wvno2);
        double drh, dm, dvdz;
        if (k == 0) { drh = drho; dm = dmu; dvdz = 0.0; }
        else { drh = drho - rho_u; dm = dmu - mu_u; dvdz = tt_top * tt_top * (1.0 / dmu - 1.0 / mu_u); }
        double dh = ut * ut * (omega2 * drh - wvno2 * dm) + dvdz;
        emit(k, db, dr, dh);
        rho_u = drho; mu_u = dmu;
        ut = ub; tt_top = tb;
    }
    SlTotals T;
    T.ugr = sumi1 / (c * sumi0);
    T.sumi1 = sumi1;
    T.fac = (0.5 / sumi1) * c / wvno2;
    return T;
}

// ---------------------------------------------------------------------------
// Earth flattening of the model.
//   swd_flatten_f32  surfdisp96.f:495-564 `sphere` (iflag 0 then 1; ar = 6370, geometry in f64, model arrays
//                    float32): what the root search sees
//   swd_bldsph       sregn96.f90:133-187 / slegn96.f90:107-167 (ar = 6371, all f64): what the eigenfunction
//                    pass sees, plus the factors vtp, dtp, rtp that map flat kernels back to the sphere
// Arrays are strided (element m at p[m * stride]).
// ---------------------------------------------------------------------------
RFS_HD void swd_flatten_f32(bool love, int n, const float* d, const float* a, const float* b, const float* rho,
                            long si, float* od, float* oa, float* ob, float* orho, long so) {
    const double ar = 6370.0;
    double dr = 0.0, r0 = ar;
    for (int i = 0; i < n; i++) {
        float di = (i == n - 1) ? 1.0f : d[i * si];
        dr = dr + (double)di;
        double r1 = ar - dr;
        double z0 = ar * log(ar / r0), z1 = ar * log(ar / r1);
        double tmp = (ar + ar) / (r0 + r1);
        od[i * so] = (i == n - 1) ? 0.0f : (float)(z1 - z0);
        oa[i * so] = (float)((double)a[i * si] * tmp);
        ob[i * so] = (float)((double)b[i * si] * tmp);
        float btp = (float)tmp, r = rho[i * si];
        if (love) { float x2 = btp * btp, x4 = x2 * x2, x5 = btp * x4; orho[i * so] = r * (1.0f / x5); }   // btp**(-5)
        else orho[i * so] = r * (float)pow((double)btp, (double)-2.275f);                                 // btp**(-2.275)
        r0 = r1;
    }
}

RFS_HD void swd_bldsph(bool love, int n, const float* d, const float* a, const float* b, const float* rho, long si,
                       double* zd, double* za, double* zb, double* zrho, double* vtp, double* dtp, double* rtp,
                       long so) {
    const double ar = 6371.0;
    double dr = 0.0, r0 = ar;
    for (int i = 0; i < n; i++) {
        dr = dr + ((i == n - 1) ? 1.0 : (double)d[i * si]);
        double r1 = ar - dr;
        double z0 = ar * log(ar / r0), z1 = ar * log(ar / r1);
        double tmp = (2.0 * ar) / (r0 + r1);
        double rt;
        if (love) { double t2 = tmp * tmp, t4 = t2 * t2; rt = 1.0 / (tmp * t4); }    // tmp**(-5)
        else rt = pow(tmp, (double)-2.275f);                                          // tmp**(-2.275)
        vtp[i * so] = tmp; rtp[i * so] = rt; dtp[i * so] = ar / r0;
        za[i * so] = (double)a[i * si] * tmp;
        zb[i * so] = (double)b[i * si] * tmp;
        zrho[i * so] = (double)rho[i * si] * rt;
        zd[i * so] = (i == n - 1) ? 0.0 : z1 - z0;
        r0 = r1;
    }
}

// ---------------------------------------------------------------------------
// Earth-flattening corrections applied to flat-model results.
//   sr_tm     sprayl / sregnpu (sregn96.f90:1625, 1851) and splove / slegnpu (slegn96.f90:661, 884):
//             tm = sqrt(1 + (k c / (2 a w))^2), k = 1 Rayleigh / 3 Love, w from the float32 pi
//   f2s_tm    _flat2sphere (surfdisp.cpp:16-49): same quantity, f64 pi, used by libsurf.forward for phase blocks
// ---------------------------------------------------------------------------
RFS_HD double sr_tm(bool love, double c, double omega) {
    double x = love ? 3.0 * c / (2. * 6371.0 * omega) : c / (2. * 6371.0 * omega);
    return sqrt(1. + x * x);
}
RFS_HD double sr_tm1(bool love, double omega, double tm) {
    double x = (love ? 1.5 : 0.5) / (6371.0 * omega);
    return x * x / tm;
}
RFS_HD double f2s_tm(bool love, double t, double c) {
    double omega = 2.0 * 3.14159265358979323846 / t;
    double x = (love ? 1.5 : 0.5) * c / (6371.0 * omega);
    return sqrt(1. + x * x);
}

}  // namespace rfs
