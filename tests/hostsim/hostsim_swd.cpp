// tests/hostsim/hostsim_swd.cpp -- TEST HARNESS ONLY (never shipped, never a fallback).
// Host build of rfsurfhmc_amd/csrc/swd_math.hpp for CPU-side checks against the oracle.
#include <vector>
#include "../../rfsurfhmc_amd/csrc/swd_math.hpp"

using namespace rfs;

extern "C" {

int hs_swd_rootsearch(int n, const float* thk, const float* vp, const float* vs, const float* rho,
                      int kmax, const double* t, double* cg, long* nsec)
{
    SwdModel M{thk, vp, vs, rho, 1, n};
    RootSearch rs;
    auto T = [&](int k) { return t[k]; };
    auto out = [&](int k, double v) { cg[k] = v; };
    rs.begin(M, T, kmax);
    while (!rs.done) {
        double del = swd_secular(M, rs.omega / rs.creq, rs.omega);
        rs.advance(del, T, out);
    }
    if (nsec) *nsec = rs.nsec;
    return rs.flag;
}

// same search driven by the split secular function (what the multi-lane GPU kernel evaluates)
int hs_swd_rootsearch_split(int n, const float* thk, const float* vp, const float* vs, const float* rho,
                            int kmax, const double* t, double* cg, long* nsec)
{
    SwdModel M{thk, vp, vs, rho, 1, n};
    std::vector<SwdLayerC> LC(n);
    for (int m = 0; m < n; m++)
        LC[m] = SwdLayerC{(double)thk[m], 1.0 / (double)vp[m], 1.0 / (double)vs[m], (double)vs[m], (double)rho[m], 1.0 / (double)rho[m]};
    RootSearch rs;
    auto T = [&](int k) { return t[k]; };
    auto out = [&](int k, double v) { cg[k] = v; };
    rs.begin(M, T, kmax);
    while (!rs.done) {
        double omega = rs.omega < 1.0e-4 ? 1.0e-4 : rs.omega, wvno = rs.omega / rs.creq;
        double wvno2 = wvno * wvno, iomega = 1.0 / omega, e[5], ent[SWD_NENT];
        swd_halfspace_e(LC[n - 1], wvno, wvno2, omega, iomega, e);
        for (int m = n - 2; m >= 0; m--) {
            swd_layer_entries(LC[m], wvno, wvno2, omega, iomega, ent);
            swd_apply_layer_raw(e, ent, -2.0 * wvno2);
            if ((m & 7) == 0) swd_rescale_pow2(e);
        }
        rs.advance(swd_finish(e), T, out);
    }
    if (nsec) *nsec = rs.nsec;
    return rs.flag;
}

// sregn96 equivalent: scaled kernels, dcdh suffix-summed; returns group velocity
double hs_sregn96(int n, const float* thk, const float* vp, const float* vs, const float* rho,
                  double t, double cp, double* dcda, double* dcdb, double* dcdh, double* dcdr)
{
    SwdModel M{thk, vp, vs, rho, 1, n};
    std::vector<double> cds(6 * n);
    double omega = 2.0 * SR_PI32 / t, wvno = omega / cp;
    sr_up(M, omega, wvno, [&](int m, const double* cd, double exe) {
        for (int i = 0; i < 5; i++) cds[6 * m + i] = cd[i];
        cds[6 * m + 5] = exe;
    });
    SrTotals T = sr_down_energy(M, omega, wvno,
        [&](int m, double* cd, double& exe) { for (int i = 0; i < 5; i++) cd[i] = cds[6 * m + i]; exe = cds[6 * m + 5]; },
        [&](int m, double da, double db, double dr, double dh) { dcda[m] = da; dcdb[m] = db; dcdr[m] = dr; dcdh[m] = dh; });
    double s = 1.0 / (T.ugr * T.sumi0);
    for (int m = 0; m < n; m++) {
        dcda[m] *= s; dcdb[m] *= s; dcdr[m] *= s;
        double dfac = T.fac * dcdh[m];
        dcdh[m] = (fabs(dfac) < 1.0e-38) ? 0.0 : dfac;
    }
    for (int i = 0; i < n - 1; i++) { double sum = 0.0; for (int j = i + 1; j < n; j++) sum += dcdh[j]; dcdh[i] = sum; }
    dcdh[n - 1] = 0.0;
    return T.ugr;
}
}
