// tests/hostsim/hostsim_swd.cpp -- TEST HARNESS ONLY (never shipped, never a fallback).
// Host build of rfsurfhmc_amd/csrc/swd_math.hpp for CPU-side checks against the oracle.
#include <vector>
#include <cstring>
#include "../../rfsurfhmc_amd/csrc/swd_math.hpp"

using namespace rfs;

// The lanes-per-item kernel's arithmetic (k_swd_roots_split) on the host: per-layer entries through the family
// interface, the recurrence either sequential or cut into nseg segments -- deepest segment from the half-space vector,
// the others from the unit vectors -- folded afterwards.  love: SwdLoveFamily instead of SwdRayFamily.
template <class F>
static double family_delta(const std::vector<SwdLayerC>& LC, int n, double omega_in, double creq, int nseg)
{
    constexpr int NENT = F::NENT, NV = F::NV;
    const double omega = omega_in < 1.0e-4 ? 1.0e-4 : omega_in, wvno = omega_in / creq, wvno2 = wvno * wvno;
    const double iomega = 1.0 / omega, tt = -2.0 * wvno2;
    std::vector<double> ent((size_t)(n - 1) * NENT);
    for (int m = 0; m < n - 1; m++) F::entries(LC[m], wvno, wvno2, omega, iomega, &ent[(size_t)m * NENT]);
    double e[NV];
    F::halfspace(LC[n - 1], wvno, wvno2, omega, iomega, e);
    if (nseg <= 1) {
        for (int m = n - 2; m >= 0; m--) {
            F::apply(e, &ent[(size_t)m * NENT], tt);
            if ((m & 7) == 0) swd_rescale_pow2_n<NV>(e);
        }
        return swd_finish_n<NV>(e);
    }
    const int seglen = (n - 1 + nseg - 1) / nseg;
    auto chain = [&](int sg, double* v) {
        int mhi = n - 2 - sg * seglen, mlo = mhi - seglen + 1;
        if (mlo < 0) mlo = 0;
        for (int m = mhi; m >= mlo; m--) F::apply(v, &ent[(size_t)m * NENT], tt);
    };
    chain(0, e);
    swd_rescale_pow2_n<NV>(e);
    for (int sg = 1; sg < nseg; sg++) {
        if (n - 2 - sg * seglen < 0) break;
        double rows[NV][NV], nw[NV];
        for (int i = 0; i < NV; i++) {
            for (int j = 0; j < NV; j++) rows[i][j] = (i == j) ? 1.0 : 0.0;
            chain(sg, rows[i]);
        }
        for (int j = 0; j < NV; j++) nw[j] = 0.0;
        for (int i = 0; i < NV; i++)
            for (int j = 0; j < NV; j++) nw[j] += e[i] * rows[i][j];
        for (int j = 0; j < NV; j++) e[j] = nw[j];
        if (sg & 1) swd_rescale_pow2_n<NV>(e);
    }
    return swd_finish_n<NV>(e);
}

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
    sr_up<true>(M, omega, wvno, [&](int m, const double* cd, double exe) {         // (<true>: a water top layer is allowed)
        for (int i = 0; i < 5; i++) cds[6 * m + i] = cd[i];
        cds[6 * m + 5] = exe;
    });
    SrTotals T = sr_down_energy<true>(M, omega, wvno,
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

int hs_rootsearch_family(int n, const float* thk, const float* vp, const float* vs, const float* rho,
                         int kmax, const double* t, double* cg, int love, int sphere, int nseg)
{
    std::vector<float> w(4 * n);
    const float *d = thk, *a = vp, *b = vs, *r = rho;
    if (sphere) {
        swd_flatten_f32(love != 0, n, thk, vp, vs, rho, 1, &w[0], &w[n], &w[2 * n], &w[3 * n], 1);
        d = &w[0]; a = &w[n]; b = &w[2 * n]; r = &w[3 * n];
    }
    SwdModel M{d, a, b, r, 1, n};
    std::vector<SwdLayerC> LC(n);
    for (int m = 0; m < n; m++)
        LC[m] = SwdLayerC{(double)d[m], 1.0 / (double)a[m], 1.0 / (double)b[m], (double)b[m], (double)r[m], 1.0 / (double)r[m]};
    RootSearch rs;
    auto T = [&](int k) { return t[k]; };
    auto out = [&](int k, double v) { cg[k] = v; };
    rs.begin(M, T, kmax);
    while (!rs.done) {
        double del = love ? family_delta<SwdLoveFamily>(LC, n, rs.omega, rs.creq, nseg)
                          : family_delta<SwdRayFamily>(LC, n, rs.omega, rs.creq, nseg);
        rs.advance(del, T, out);
    }
    return rs.flag;
}

// Love / spherical-earth variants --------------------------------------------------------------
// root search on the (optionally earth-flattened) float32 model; love: dltar1 instead of dltar4
int hs_rootsearch_general(int n, const float* thk, const float* vp, const float* vs, const float* rho,
                          int kmax, const double* t, double* cg, int love, int sphere)
{
    std::vector<float> w(4 * n);
    const float *d = thk, *a = vp, *b = vs, *r = rho;
    if (sphere) {
        swd_flatten_f32(love != 0, n, thk, vp, vs, rho, 1, &w[0], &w[n], &w[2 * n], &w[3 * n], 1);
        d = &w[0]; a = &w[n]; b = &w[2 * n]; r = &w[3 * n];
    }
    SwdModel M{d, a, b, r, 1, n};
    RootSearch rs;
    auto T = [&](int k) { return t[k]; };
    auto out = [&](int k, double v) { cg[k] = v; };
    rs.begin(M, T, kmax);
    while (!rs.done) {
        double wvno = rs.omega / rs.creq;
        double del = love ? swd_secular_love(M, wvno, rs.omega) : swd_secular(M, wvno, rs.omega);
        rs.advance(del, T, out);
    }
    return rs.flag;
}

}  // extern "C"

template <class Mdl>
static double rayleigh_flat_kernels(const Mdl& M, int n, double omega, double wvno, double* dcda, double* dcdb,
                                    double* dcdh, double* dcdr)
{
    std::vector<double> cds(6 * n);
    sr_up<true>(M, omega, wvno, [&](int m, const double* cd, double exe) {         // (<true>: a water top layer is allowed)
        for (int i = 0; i < 5; i++) cds[6 * m + i] = cd[i];
        cds[6 * m + 5] = exe;
    });
    SrTotals T = sr_down_energy<true>(M, omega, wvno,
        [&](int m, double* cd, double& exe) { for (int i = 0; i < 5; i++) cd[i] = cds[6 * m + i]; exe = cds[6 * m + 5]; },
        [&](int m, double da, double db, double dr, double dh) { dcda[m] = da; dcdb[m] = db; dcdr[m] = dr; dcdh[m] = dh; });
    double s = 1.0 / (T.ugr * T.sumi0);
    for (int m = 0; m < n; m++) {
        dcda[m] *= s; dcdb[m] *= s; dcdr[m] *= s;
        double dfac = T.fac * dcdh[m];
        dcdh[m] = (fabs(dfac) < 1.0e-38) ? 0.0 : dfac;
    }
    return T.ugr;
}

template <class Mdl>
static double love_flat_kernels(const Mdl& M, int n, double omega, double wvno, double* dcdb, double* dcdh, double* dcdr)
{
    std::vector<double> sc(3 * n);
    sl_up(M, omega, wvno, [&](int m, double uu, double tt, double exl) { sc[3 * m] = uu; sc[3 * m + 1] = tt; sc[3 * m + 2] = exl; });
    SlTotals T = sl_down_energy(M, omega, wvno,
        [&](int m, double& uu, double& tt, double& exl) { uu = sc[3 * m]; tt = sc[3 * m + 1]; exl = sc[3 * m + 2]; },
        [&](int m, double db, double dr, double dh) { dcdb[m] = db; dcdr[m] = dr; dcdh[m] = dh; });
    for (int m = 0; m < n; m++) {
        dcdb[m] /= T.sumi1; dcdr[m] /= T.sumi1;
        double dfac = T.fac * dcdh[m];
        dcdh[m] = (fabs(dfac) < 1.0e-38) ? 0.0 : dfac;
    }
    return T.ugr;
}

extern "C" {
// sregn96 / slegn96 with iflsph: kernels of the (possibly flattened) model mapped back to the sphere, thickness
// kernels suffix-summed; *cp in = flat phase velocity, out = spherical; returns U (spherical)
double hs_eigen_general(int n, const float* thk, const float* vp, const float* vs, const float* rho, double t,
                        double* cp, double* dcda, double* dcdb, double* dcdh, double* dcdr, int love, int sphere)
{
    double omega = 2.0 * SR_PI32 / t, wvno = omega / *cp, u;
    std::vector<double> z(7 * n);
    for (int m = 0; m < n; m++) dcda[m] = 0.0;
    if (sphere) {
        swd_bldsph(love != 0, n, thk, vp, vs, rho, 1, &z[0], &z[n], &z[2 * n], &z[3 * n], &z[4 * n], &z[5 * n], &z[6 * n], 1);
        SwdModelD M{&z[0], &z[n], &z[2 * n], &z[3 * n], 1, n};
        u = love ? love_flat_kernels(M, n, omega, wvno, dcdb, dcdh, dcdr)
                 : rayleigh_flat_kernels(M, n, omega, wvno, dcda, dcdb, dcdh, dcdr);
        double tm = sr_tm(love != 0, *cp, omega), tm3 = tm * tm * tm;
        for (int m = 0; m < n; m++) {
            dcda[m] = dcda[m] * z[4 * n + m] / tm3; dcdb[m] = dcdb[m] * z[4 * n + m] / tm3;
            dcdh[m] = dcdh[m] * z[5 * n + m] / tm3; dcdr[m] = dcdr[m] * z[6 * n + m] / tm3;
        }
        *cp = *cp / tm; u = u * tm;
    } else {
        SwdModel M{thk, vp, vs, rho, 1, n};
        u = love ? love_flat_kernels(M, n, omega, wvno, dcdb, dcdh, dcdr)
                 : rayleigh_flat_kernels(M, n, omega, wvno, dcda, dcdb, dcdh, dcdr);
    }
    for (int i = 0; i < n - 1; i++) { double sum = 0.0; for (int j = i + 1; j < n; j++) sum += dcdh[j]; dcdh[i] = sum; }
    dcdh[n - 1] = 0.0;
    return u;
}
}

// A warm search that changes lanes every `budget` evaluations, as in k_swd_warm's rounds: its 12 doubles and the packed word
// (WarmSearch::pack_small) are all that travels; the receiving machine starts from garbage.  cout / nev / status as
// hs_warm_roots; the caller compares them with the uninterrupted search.  Returns the number of hand-overs made.
extern "C" int hs_warm_roots_handover(int n, const float* thk, const float* vp, const float* vs, const float* rho, int nt,
                                      const double* t, const double* cprev, const double* dc, const double* l1, int budget,
                                      double* cout, int* nev, int* status)
{
    std::vector<SwdLayerC> LC(n);
    float betmx = -1.e20f;
    for (int m = 0; m < n; m++) {
        LC[m] = SwdLayerC{(double)thk[m], 1.0 / (double)vp[m], 1.0 / (double)vs[m], (double)vs[m], (double)rho[m], 1.0 / (double)rho[m]};
        if (vs[m] > betmx) betmx = vs[m];
    }
    auto loadL = [&](int m) { return LC[m]; };
    int moves = 0;
    for (int k = 0; k < nt; k++) {
        const double omega = (2.0 * 3.141592653589793) / t[k];
        WarmSearch ws;
        ws.begin(cprev[k], dc[k], l1[k]);
        int left = budget, attempt = 0, nev_first = 0;
        while (ws.active()) {
            ws.advance(swd_secular_family<SwdRayFamily>(n, loadL, omega, ws.creq));
            if (--left == 0 && ws.active()) {
                const double d[12] = {ws.cpred, ws.eps, ws.R, ws.a, ws.fa, ws.b, ws.fb, ws.creq, ws.root, ws.slope, ws.f0, ws.mlast};
                const unsigned long long bits = ws.pack_small(attempt, nev_first);
                WarmSearch w2;
                memset((void*)&w2, 0xA5, sizeof(w2));
                w2.cpred = d[0]; w2.eps = d[1]; w2.R = d[2]; w2.a = d[3]; w2.fa = d[4]; w2.b = d[5]; w2.fb = d[6];
                w2.creq = d[7]; w2.root = d[8]; w2.slope = d[9]; w2.f0 = d[10]; w2.mlast = d[11];
                w2.unpack_small(bits, attempt, nev_first);
                ws = w2; left = budget; moves++;
            }
        }
        const bool ok = ws.phase == WarmSearch::W_DONE && !(ws.root > (double)betmx);
        cout[k] = ok ? (double)(float)ws.root : 0.0;
        nev[k] = ws.nev; status[k] = ok ? 1 : 0;
    }
    return moves;
}

// Warm-started refinement (WarmSearch + swd_secular_family: the lane code of k_swd_warm) for the nt periods of one
// model: cprev / dc / l1 per period as the kernel's predictor hands them over.  status[k] = 1 accepted, 0 declined;
// nev[k] = secular evaluations spent.  Returns the number of declined periods.
extern "C" int hs_warm_roots(int n, const float* thk, const float* vp, const float* vs, const float* rho, int nt,
                             const double* t, const double* cprev, const double* dc, const double* l1, int love,
                             int sphere, double* cout, int* nev, int* status)
{
    std::vector<float> w(4 * n);
    const float *d = thk, *a = vp, *b = vs, *r = rho;
    if (sphere) {
        swd_flatten_f32(love != 0, n, thk, vp, vs, rho, 1, &w[0], &w[n], &w[2 * n], &w[3 * n], 1);
        d = &w[0]; a = &w[n]; b = &w[2 * n]; r = &w[3 * n];
    }
    std::vector<SwdLayerC> LC(n);
    float betmx = -1.e20f;
    for (int m = 0; m < n; m++) {
        LC[m] = SwdLayerC{(double)d[m], 1.0 / (double)a[m], 1.0 / (double)b[m], (double)b[m], (double)r[m], 1.0 / (double)r[m]};
        if (b[m] > betmx) betmx = b[m];
    }
    auto loadL = [&](int m) { return LC[m]; };
    int nfail = 0;
    std::vector<int> sgn_lo(nt, 0);
    for (int k = 0; k < nt; k++) {
        const double omega = (2.0 * 3.141592653589793) / t[k];
        WarmSearch ws;
        ws.begin(cprev[k], dc[k], l1[k]);
        while (ws.active()) {
            double f = love ? swd_secular_family<SwdLoveFamily>(n, loadL, omega, ws.creq)
                            : swd_secular_family<SwdRayFamily>(n, loadL, omega, ws.creq);
            ws.advance(f);
        }
        bool ok = ws.phase == WarmSearch::W_DONE && !(ws.root > (double)betmx);
        cout[k] = ok ? (double)(float)ws.root : 0.0;
        nev[k] = ws.nev; status[k] = ok ? 1 : 0;
        sgn_lo[k] = signbit(ws.fa) ? 1 : 0;
    }
    // the branch test of k_swd_warm_check: regular sequence -> one evaluation at the scan's start point; irregular -> the
    // reference's own scan grid is walked from there
    SwdModel M{d, a, b, r, 1, n};
    float bmx;
    const double cc = (double)swd_start_value(M, bmx), dcs = (double)0.005f;
    bool irregular = false, all_ok = true;
    for (int k = 0; k < nt; k++) all_ok = all_ok && status[k] == 1;
    for (int k = 1; k < nt && all_ok; k++) irregular = irregular || (cout[k - 1] - 1.5 * dcs >= cout[k]);
    for (int k = 0; k < nt; k++) irregular = irregular || l1[k] > WARM_L1MAX;          // wide moves walk the grid too
    for (int k = 0; k < nt && all_ok; k++) {
        const double omega = (2.0 * 3.141592653589793) / t[k];
        auto sec = [&](double w, double c) { return love ? swd_secular_family<SwdLoveFamily>(n, loadL, w, c)
                                                          : swd_secular_family<SwdRayFamily>(n, loadL, w, c); };
        const double ck = cout[k], sk = k == 0 ? cc : cout[k - 1] - 1.5 * dcs;
        if (!(sk > 0.0) || sk == ck) { status[k] = 2; continue; }
        double f = sec(omega, sk);
        nev[k]++;
        if (!irregular) { if (!(sk < ck) || (signbit(f) ? 1 : 0) != sgn_lo[k]) status[k] = 2; continue; }
        int s1st = signbit(f) ? 1 : 0;
        if (k > 0) { s1st = signbit(sec((2.0 * 3.141592653589793) / t[0], cc)) ? 1 : 0; nev[k]++; }
        int idir = (k == 0 || (signbit(f) ? 1 : 0) == s1st) ? +1 : -1;
        double c1 = sk;
        for (int steps = 0;; steps++) {
            double c2 = idir > 0 ? c1 + dcs : c1 - dcs;
            if (c2 <= cc || steps > (k == 0 ? 1024 : 400)) { status[k] = 2; break; }
            const double f2 = sec(omega, c2);
            nev[k]++;
            if (diffsign(f, f2)) { if (!(fmin(c1, c2) < ck && ck < fmax(c1, c2))) status[k] = 2; break; }
            c1 = c2; f = f2;
            if (c1 < cc || c1 >= (double)bmx + dcs) { status[k] = 2; break; }
        }
    }
    for (int k = 0; k < nt; k++) nfail += status[k] != 1;
    return nfail;
}

// The mode loop (RootSearchModes: surfdisp96.f:227-316 + the per-period retry of surfdisp.cpp:93-100) on the
// (optionally earth-flattened) float32 model: `mode` = libsurf's argument (0 fundamental, 1 first higher mode, ...).
extern "C" int hs_rootsearch_modes(int n, const float* thk, const float* vp, const float* vs, const float* rho,
                                   int kmax, const double* t, double* cg, int love, int sphere, int mode)
{
    std::vector<float> w(4 * n);
    const float *d = thk, *a = vp, *b = vs, *r = rho;
    if (sphere) {
        swd_flatten_f32(love != 0, n, thk, vp, vs, rho, 1, &w[0], &w[n], &w[2 * n], &w[3 * n], 1);
        d = &w[0]; a = &w[n]; b = &w[2 * n]; r = &w[3 * n];
    }
    SwdModel M{d, a, b, r, 1, n};
    std::vector<double> craw(kmax, 0.0);
    struct Out { double* cg; void operator()(int k, double v) const { cg[k] = v; } double get(int k) const { return cg[k]; } };
    Out out{cg};
    for (int k = 0; k < kmax; k++) cg[k] = 0.0;
    RootSearchModes rs;
    auto T = [&](int k) { return t[k]; };
    rs.set_modes(mode + 1, craw.data(), 1);
    rs.begin(M, T, kmax);
    while (!rs.done) {
        double wvno = rs.omega / rs.creq;
        double del = love ? swd_secular_love(M, wvno, rs.omega) : swd_secular(M, wvno, rs.omega);
        rs.advance(del, T, out);
    }
    return rs.flag;
}

// the plain secular functions, one evaluation (experiments and checks of derivative identities)
extern "C" double hs_secular(int n, const float* thk, const float* vp, const float* vs, const float* rho, double omega, double c, int love)
{
    SwdModel M{thk, vp, vs, rho, 1, n};
    return love ? swd_secular_love(M, omega / c, omega) : swd_secular(M, omega / c, omega);
}

// The reference's roots from approximate ones (ExactGroup, swd_math.hpp: the lane code of k_swd_exact): the nt periods of one
// sequence in groups of `G` with `runup` run-up periods each; approx[k] = warm-started root of period k (float32-rounded,
// as k_swd_warm leaves it).  cout[k] = float32-rounded result (0 where the group declined), status[k] = 1 / 0,
// nev[group] = secular evaluations.  lazy = 0: every value nevill asks for is evaluated (CellNevillT<false>).
// Returns the number of declined groups.
template <bool LAZY>
static int exact_roots(int n, const float* thk, const float* vp, const float* vs, const float* rho, int nt,
                       const double* t, const double* approx, int love, int sphere, int G, int runup,
                       double* cout, int* status, int* nev, int* cause, long* nsupplied)
{
    std::vector<float> w(4 * n);
    const float *d = thk, *a = vp, *b = vs, *r = rho;
    if (sphere) {
        swd_flatten_f32(love != 0, n, thk, vp, vs, rho, 1, &w[0], &w[n], &w[2 * n], &w[3 * n], 1);
        d = &w[0]; a = &w[n]; b = &w[2 * n]; r = &w[3 * n];
    }
    std::vector<SwdLayerC> LC(n);
    for (int m = 0; m < n; m++)
        LC[m] = SwdLayerC{(double)d[m], 1.0 / (double)a[m], 1.0 / (double)b[m], (double)b[m], (double)r[m], 1.0 / (double)r[m]};
    auto loadL = [&](int m) { return LC[m]; };
    SwdModel M{d, a, b, r, 1, n};
    float bmx;
    const double cc = (double)swd_start_value(M, bmx);
    auto om = [&](int k) { return (2.0 * 3.141592653589793) / t[k]; };
    auto ap = [&](int k) { return approx[k]; };
    int nfail = 0, g = 0;
    for (int k = 0; k < nt; k++) { cout[k] = 0.0; status[k] = 0; }
    for (int k0 = 0; k0 < nt; k0 += G, g++) {
        const int k1 = k0 + G < nt ? k0 + G : nt, kr = k0 - runup > 0 ? k0 - runup : 0;
        ExactGroupT<LAZY> x;
        double tab[24];
        x.begin(kr, k0, k1, cc, bmx, kr > 0 ? approx[kr - 1] * (1.0 - EXACT_OFFSET) : 0.0, ap, om, tab, 1);
        bool fin = false;
        while (x.active() && !fin) {
            const double f = love ? swd_secular_family<SwdLoveFamily>(n, loadL, x.omega, x.creq)
                                  : swd_secular_family<SwdRayFamily>(n, loadL, x.omega, x.creq);
            x.advance(f);
            if (x.phase == ExactGroupT<LAZY>::X_DONE) {
                if (x.wanted()) { cout[x.k] = (double)(float)x.root(); status[x.k] = 1; }
                if (!x.next(ap, om)) fin = true;
            }
        }
        nev[g] = x.nev; cause[g] = x.phase == ExactGroupT<LAZY>::X_FAIL ? x.cause : 0;
        if (nsupplied) *nsupplied += x.nsupplied;
        if (x.phase == ExactGroupT<LAZY>::X_FAIL) { nfail++; for (int k = k0; k < k1; k++) { status[k] = 0; cout[k] = 0.0; } }
    }
    return nfail;
}
// The same with every group's machine SAVED and LOADED (ExactGroupT::save / load, the hand-over of k_swd_exact in rounds) after
// every `budget` evaluations -- into a fresh machine with its own table storage.  Returns the number of hand-overs.
extern "C" int hs_exact_roots_handover(int n, const float* thk, const float* vp, const float* vs, const float* rho, int nt,
                                       const double* t, const double* approx, int G, int runup, int budget,
                                       double* cout, int* status, int* nev)
{
    std::vector<SwdLayerC> LC(n);
    for (int m = 0; m < n; m++)
        LC[m] = SwdLayerC{(double)thk[m], 1.0 / (double)vp[m], 1.0 / (double)vs[m], (double)vs[m], (double)rho[m], 1.0 / (double)rho[m]};
    auto loadL = [&](int m) { return LC[m]; };
    SwdModel M{thk, vp, vs, rho, 1, n};
    float bmx;
    const double cc = (double)swd_start_value(M, bmx);
    auto om = [&](int k) { return (2.0 * 3.141592653589793) / t[k]; };
    auto ap = [&](int k) { return approx[k]; };
    int moves = 0, g = 0;
    for (int k = 0; k < nt; k++) { cout[k] = 0.0; status[k] = 0; }
    for (int k0 = 0; k0 < nt; k0 += G, g++) {
        const int k1 = k0 + G < nt ? k0 + G : nt, kr = k0 - runup > 0 ? k0 - runup : 0;
        ExactGroup* x = new ExactGroup;
        double* tab = new double[24 * 3];                            // (stride 3: the table's stride travels with load())
        x->begin(kr, k0, k1, cc, bmx, kr > 0 ? approx[kr - 1] * (1.0 - EXACT_OFFSET) : 0.0, ap, om, tab, 3);
        bool fin = false;
        int left = budget;
        while (x->active() && !fin) {
            x->advance(swd_secular_family<SwdRayFamily>(n, loadL, x->omega, x->creq));
            if (x->phase == ExactGroup::X_DONE) {
                if (x->wanted()) { cout[x->k] = (double)(float)x->root(); status[x->k] = 1; }
                if (!x->next(ap, om)) fin = true;
            }
            if (--left == 0 && x->active() && !fin) {               // hand the machine on
                std::vector<double> D(EXACT_SPILL_ND * 5, -7.0);
                x->save(&D[2], 5);                                   // (slot 2 of a list of 5)
                std::memset((void*)x, 0xA5, sizeof(ExactGroup)); delete x; delete[] tab;
                x = new ExactGroup; tab = new double[24];
                x->load(&D[2], 5, tab, 1);
                left = budget; moves++;
            }
        }
        nev[g] = x->nev;
        if (x->phase == ExactGroup::X_FAIL) for (int k = k0; k < k1; k++) { status[k] = 0; cout[k] = 0.0; }
        delete x; delete[] tab;
    }
    return moves;
}
extern "C" int hs_exact_roots(int n, const float* thk, const float* vp, const float* vs, const float* rho, int nt,
                              const double* t, const double* approx, int love, int sphere, int G, int runup,
                              double* cout, int* status, int* nev, int* cause)
{
    return exact_roots<true>(n, thk, vp, vs, rho, nt, t, approx, love, sphere, G, runup, cout, status, nev, cause, nullptr);
}
extern "C" int hs_exact_roots2(int n, const float* thk, const float* vp, const float* vs, const float* rho, int nt,
                               const double* t, const double* approx, int love, int sphere, int G, int runup, int lazy,
                               double* cout, int* status, int* nev, int* cause, long* nsupplied)
{
    *nsupplied = 0;
    return lazy ? exact_roots<true>(n, thk, vp, vs, rho, nt, t, approx, love, sphere, G, runup, cout, status, nev, cause, nsupplied)
                : exact_roots<false>(n, thk, vp, vs, rho, nt, t, approx, love, sphere, G, runup, cout, status, nev, cause, nsupplied);
}

// The table of k_swd_cold_scan (round 6) with the device's own secular function and start value: out[k * np + i] = the Rayleigh /
// Love secular function of period t[k] at c0 + i dc; hs_start_value: the model's start value and fastest S velocity.
extern "C" double hs_start_value(int n, const float* thk, const float* vp, const float* vs, const float* rho, float* bmx)
{
    SwdModel M{thk, vp, vs, rho, 1, n};
    float b = 0.f;
    const float cc = swd_start_value(M, b);
    *bmx = b;
    return (double)cc;
}
extern "C" void hs_secular_table(int n, const float* thk, const float* vp, const float* vs, const float* rho, int nper, const double* t,
                                 int np, double c0, double dc, int love, double* out)
{
    SwdModel M{thk, vp, vs, rho, 1, n};
    for (int k = 0; k < nper; k++) {
        const double omega = (2.0 * 3.141592653589793) / t[k];
        for (int i = 0; i < np; i++) {
            const double c = c0 + (double)i * dc;
            out[(size_t)k * np + i] = love ? swd_secular_love(M, omega / c, omega) : swd_secular(M, omega / c, omega);
        }
    }
}
