// tests/hostsim/hostsim_rf.cpp -- TEST HARNESS ONLY (never shipped, never a fallback).
// Compiles the device math headers of rfsurfhmc_amd/csrc for the host with g++ so the
// lane-level algorithms can be checked against the oracle in the CPU-only container.
// The product path is the HIP library; nothing in rfsurfhmc_amd/ loads this file.
#include <vector>
#include "../../rfsurfhmc_amd/csrc/rf_math.hpp"

using namespace rfs;

extern "C" {

// One frequency: R21, R22 and the reference-shaped partials R21_m/R22_m [4][n] (complex as
// interleaved doubles), computed with the O(n) row/column sweeps of rf_math.hpp.
void hs_rf_response_par_all(int n, const double* thk, const double* rho, const double* vp,
                            const double* vs, const double* qa, const double* qb, double p,
                            double w_re, double w_im, int rf_type, double* R21, double* R22,
                            double* R21m, double* R22m)
{
    std::vector<RfLayer> L(n);
    for (int j = 0; j < n; j++) rf_make_layer(L[j], thk[j], rho[j], vp[j], vs[j], qa[j], qb[j], p);
    cplx omega = C(w_re, w_im), k = omega * p;
    std::vector<V4> rs(n);
    V4 r = rf_einv_row(L[n - 1], rf_type);
    for (int j = n - 2; j >= 0; j--) {
        rs[j] = r;
        RfHyp H; RfA A;
        rf_hyp(L[j], omega, H);
        rf_build_A(L[j], H, A);
        r = rf_row_times_A(r, A);
    }
    int c21 = (rf_type == 1) ? 0 : 1, c22 = (rf_type == 1) ? 1 : 0;
    cplx r21 = r.v[c21];
    cplx r22 = (rf_type == 1) ? mul_i(r.v[c22]) : -mul_i(r.v[c22]);
    R21[0] = r21.re; R21[1] = r21.im; R22[0] = r22.re; R22[1] = r22.im;
    for (int which = 0; which < 2; which++) {
        V4 y; for (int i = 0; i < 4; i++) y.v[i] = C(0.0);
        y.v[which == 0 ? c21 : c22] = C(1.0);
        double* out = which == 0 ? R21m : R22m;
        for (int j = 0; j < n; j++) {
            cplx T[4];
            if (j < n - 1) {
                RfHyp H; RfA A;
                rf_hyp(L[j], omega, H);
                rf_layer_partials(L[j], H, k, rs[j], y, T);
                rf_build_A(L[j], H, A);
                y = rf_A_times_col(A, y);
            } else {
                rf_half_partials(L[j], omega, rf_type, y, T);
            }
            for (int ip = 0; ip < 4; ip++) {
                cplx t = T[ip];
                if (which == 1) t = (rf_type == 1) ? mul_i(t) : -mul_i(t);
                out[2 * (ip * n + j)] = t.re; out[2 * (ip * n + j) + 1] = t.im;
            }
        }
    }
}

// Row peeling (k_rf_passB<., true>): the stored rows of the bottom-up sweep against the rows rebuilt from the final row
// with rf_row_times_Ainv, and the density partial from the commutator (rf_rho_partial) against the closed form, one
// frequency.  out[0] = worst relative row error, out[1] = worst relative error of Re(T_rho) against max |Re(T_rho)|.
void hs_rf_peeling_errors(int n, const double* thk, const double* rho, const double* vp, const double* vs,
                          const double* qa, const double* qb, double p, double w_re, double w_im, int rf_type, double* out)
{
    std::vector<RfLayer> L(n);
    for (int j = 0; j < n; j++) rf_make_layer(L[j], thk[j], rho[j], vp[j], vs[j], qa[j], qb[j], p);
    cplx omega = C(w_re, w_im), k = omega * p;
    std::vector<V4> rs(n);
    V4 r = rf_einv_row(L[n - 1], rf_type);
    for (int j = n - 2; j >= 0; j--) {
        rs[j] = r;
        RfHyp H; RfA A;
        rf_hyp(L[j], omega, H); rf_build_A(L[j], H, A);
        r = rf_row_times_A(r, A);
    }
    V4 y; for (int i = 0; i < 4; i++) y.v[i] = C(0.0);
    y.v[0] = C(0.3, -0.2); y.v[1] = C(-0.1, 0.7);
    double erow = 0.0, etr = 0.0, tmax = 0.0;
    std::vector<double> ta(n), tb(n);
    for (int j = 0; j < n - 1; j++) {
        RfHyp H; RfA A;
        rf_hyp(L[j], omega, H); rf_build_A(L[j], H, A);
        const V4 ra = r;
        r = rf_row_times_Ainv(r, A);
        double d = 0.0, m = 0.0;
        for (int i = 0; i < 4; i++) { d += norm2(r.v[i] - rs[j].v[i]); m += norm2(rs[j].v[i]); }
        if (sqrt(d / m) > erow) erow = sqrt(d / m);
        cplx T[4];
        rf_layer_partials(L[j], H, k, rs[j], y, T);
        const V4 ya = rf_A_times_col(A, y);
        ta[j] = T[0].re; tb[j] = rf_rho_partial(L[j], ra, y, r, ya);
        if (fabs(ta[j]) > tmax) tmax = fabs(ta[j]);
        y = ya;
    }
    for (int j = 0; j < n - 1; j++) if (fabs(ta[j] - tb[j]) / tmax > etr) etr = fabs(ta[j] - tb[j]) / tmax;
    out[0] = erow; out[1] = etr;
}

// The float32 step of pass A's sweep beyond the band (rf_row_step_f32) against the f64 sweep, one frequency:
// out[0] = relative error of |R21|^2, out[1] = the growth exponent the kernels would compute (rf_growth_exponent),
// out[2] = the largest exponent the float32 sweep is allowed at this layer count (rf_f32_emax).
void hs_rf_f32_error(int n, const double* thk, const double* rho, const double* vp, const double* vs,
                     const double* qa, const double* qb, double p, double w_re, double w_im, int rf_type, double* out)
{
    std::vector<RfLayer> L(n);
    for (int j = 0; j < n; j++) rf_make_layer(L[j], thk[j], rho[j], vp[j], vs[j], qa[j], qb[j], p);
    cplx omega = C(w_re, w_im);
    V4 r = rf_einv_row(L[n - 1], rf_type);
    V4f q;
    for (int i = 0; i < 4; i++) q.v[i] = to_f32(r.v[i]);
    for (int j = n - 2; j >= 0; j--) {
        RfHyp H; RfA A;
        rf_hyp(L[j], omega, H); rf_build_A(L[j], H, A);
        r = rf_row_times_A(r, A);
        q = rf_row_step_f32(L[j], omega, q);
    }
    const int c21 = (rf_type == 1) ? 0 : 1;
    const double a = norm2(r.v[c21]);
    const double b = (double)cf_re(q.v[c21]) * (double)cf_re(q.v[c21]) + (double)cf_im(q.v[c21]) * (double)cf_im(q.v[c21]);
    out[0] = fabs(b / a - 1.0);
    out[1] = rf_growth_exponent(L.data(), n, -w_im, w_re);
    out[2] = rf_f32_emax(n);
}

int hs_rf_f32_decide(double water, double b1, double b2, double h1, double h2, double lo1, double lo2)
{
    return rf_f32_decide(water, b1, b2, h1, h2, lo1, lo2);
}
double hs_rf_f32_margin() { return RF_F32_MARGIN; }
}
