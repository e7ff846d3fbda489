/*
 * oracle/swd_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's Rayleigh-wave dispersion path
 * (no water layer: the water branches are checked against fixtures of the compiled reference
 * instead, tests/test_water.py), one function per reference
 * routine, each citing the file:line it follows under /root/reference:
 *
 *   src/SWD/surfdisp96.f   root search  (surfdisp96, gtsolh, getsol, nevill,
 *                          half, dltar4, var, dnka, normc)
 *   src/SWD/sregn96.f90    eigenfunctions + Frechet kernels (sregn96, sregnpu,
 *                          svfunc, up, down, dnka, hska, evalg, varsv, energy,
 *                          intijr, ffunc/gfunc/h1func/h2func, getdcdh, getmat)
 *   src/SWD/surfdisp.cpp   per-period glue (_surfdisp, _RayleighGroup, _SurfKernel)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this file's shared object; the product (rfsurfhmc_amd/) never does.  It is
 * pinned against the reference itself (oracle/_ref/libsurf*.so built from the
 * reference sources, see oracle/Makefile) by tests/test_oracle_vs_ref.py and
 * against the committed golden vectors by tests/test_oracle_golden.py.
 *
 * Precision quirks that are reproduced on purpose (they move results at the
 * 1e-7 level): float32 model arrays, float32 start value (gtsolh), float32
 * search increment 0.005f, float32-rounded phase velocity on output,
 * float32 pi in sregn96, float32 rho*rho and 2*b*b inside the sregn96
 * compound/Haskell matrices, the single-precision literal 0.01 in nevill.
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef double _Complex zc;

#define ORC_MAXL 512

static double dsign1(double x) { return copysign(1.0, x); }

/* ------------------------------------------------------------------ */
/* surfdisp96.f:1015-1040  normc (5-vector)                            */
static void sd_normc(double *ee, double *ex)
{
    double t1 = 0.0;
    for (int i = 0; i < 5; i++)
        if (fabs(ee[i]) > t1) t1 = fabs(ee[i]);
    if (t1 < 1.0e-40) t1 = 1.0;
    for (int i = 0; i < 5; i++) ee[i] = ee[i] / t1;
    *ex = log(t1);
}

typedef struct {
    double a0, cpcq, cpy, cpz, cqw, cqx, xy, xz, wy, wz;
} sd_prod;

/* surfdisp96.f:894-1011  var */
static void sd_var(double p, double q, double ra, double rb, double wvno,
                   double xka, double xkb, double dpth, double *w_out,
                   double *cosp_out, double *exa_out, sd_prod *o)
{
    double pex = 0.0, sex = 0.0, cosp = 0, sinp, w = 0, x = 0, fac;
    double cosq = 0, sinq, y = 0, z = 0;
    if (wvno < xka) {
        sinp = sin(p); w = sinp / ra; x = -ra * sinp; cosp = cos(p);
    } else if (wvno == xka) {
        cosp = 1.0; w = dpth; x = 0.0;
    } else {
        pex = p; fac = 0.0;
        if (p < 16) fac = exp(-2.0 * p);
        cosp = (1.0 + fac) * 0.5; sinp = (1.0 - fac) * 0.5;
        w = sinp / ra; x = ra * sinp;
    }
    if (wvno < xkb) {
        sinq = sin(q); y = sinq / rb; z = -rb * sinq; cosq = cos(q);
    } else if (wvno == xkb) {
        cosq = 1.0; y = dpth; z = 0.0;
    } else {
        sex = q; fac = 0.0;
        if (q < 16) fac = exp(-2.0 * q);
        cosq = (1.0 + fac) * 0.5; sinq = (1.0 - fac) * 0.5;
        y = sinq / rb; z = rb * sinq;
    }
    double exa = pex + sex;
    o->a0 = 0.0;
    if (exa < 60.0) o->a0 = exp(-exa);
    o->cpcq = cosp * cosq; o->cpy = cosp * y; o->cpz = cosp * z;
    o->cqw = cosq * w; o->cqx = cosq * x;
    o->xy = x * y; o->xz = x * z; o->wy = w * y; o->wz = w * z;
    *w_out = w; *cosp_out = cosp; *exa_out = exa;
}

/* surfdisp96.f:1044-1088  dnka (Dunkin 5x5, ca[row][col], 0-based) */
static void sd_dnka(double ca[5][5], double wvno2, double gam, double gammk,
                    double rho, const sd_prod *v)
{
    const double one = 1.0, two = 2.0;
    double gamm1 = gam - one, twgm1 = gam + gamm1, gmgmk = gam * gammk;
    double gmgm1 = gam * gamm1, gm1sq = gamm1 * gamm1, rho2 = rho * rho;
    double a0pq = v->a0 - v->cpcq;
    ca[0][0] = v->cpcq - two * gmgm1 * a0pq - gmgmk * v->xz - wvno2 * gm1sq * v->wy;
    ca[0][1] = (wvno2 * v->cpy - v->cqx) / rho;
    ca[0][2] = -(twgm1 * a0pq + gammk * v->xz + wvno2 * gamm1 * v->wy) / rho;
    ca[0][3] = (v->cpz - wvno2 * v->cqw) / rho;
    ca[0][4] = -(two * wvno2 * a0pq + v->xz + wvno2 * wvno2 * v->wy) / rho2;
    ca[1][0] = (gmgmk * v->cpz - gm1sq * v->cqw) * rho;
    ca[1][1] = v->cpcq;
    ca[1][2] = gammk * v->cpz - gamm1 * v->cqw;
    ca[1][3] = -v->wz;
    ca[1][4] = ca[0][3];
    ca[3][0] = (gm1sq * v->cpy - gmgmk * v->cqx) * rho;
    ca[3][1] = -v->xy;
    ca[3][2] = gamm1 * v->cpy - gammk * v->cqx;
    ca[3][3] = ca[1][1];
    ca[3][4] = ca[0][1];
    ca[4][0] = -(two * gmgmk * gm1sq * a0pq + gmgmk * gmgmk * v->xz +
                 gm1sq * gm1sq * v->wy) * rho2;
    ca[4][1] = ca[3][0];
    ca[4][2] = -(gammk * gamm1 * twgm1 * a0pq + gam * gammk * gammk * v->xz +
                 gamm1 * gm1sq * v->wy) * rho;
    ca[4][3] = ca[1][0];
    ca[4][4] = ca[0][0];
    double t = -two * wvno2;
    ca[2][0] = t * ca[4][2];
    ca[2][1] = t * ca[3][2];
    ca[2][2] = v->a0 + two * (v->cpcq - ca[0][0]);
    ca[2][3] = t * ca[1][2];
    ca[2][4] = t * ca[0][2];
}

typedef struct {
    int n;
    const float *d, *a, *b, *rho;
    long nsec; /* secular-function evaluation counter (for work accounting) */
    int ifunc; /* 1 Love (dltar1), 2 Rayleigh (dltar4): surfdisp96.f:705-723 */
} sd_model;

/* surfdisp96.f:727-787  dltar1 (Love, Haskell-Thompson from the half-space up; llw = 1) */
static double sd_dltar1(double wvno, double omega, sd_model *M)
{
    int mmax = M->n;
    M->nsec++;
    double beta1 = (double)M->b[mmax - 1], rho1 = (double)M->rho[mmax - 1];
    double xkb = omega / beta1;
    double wvnop = wvno + xkb, wvnom = fabs(wvno - xkb);
    double rb = sqrt(wvnop * wvnom);
    double e1 = rho1 * rb, e2 = 1.0 / (beta1 * beta1);
    for (int m = mmax - 2; m >= 0; m--) {
        beta1 = (double)M->b[m]; rho1 = (double)M->rho[m];
        double xmu = rho1 * beta1 * beta1;
        xkb = omega / beta1;
        wvnop = wvno + xkb; wvnom = fabs(wvno - xkb);
        rb = sqrt(wvnop * wvnom);
        double q = (double)M->d[m] * rb, y, z, cosq;
        if (wvno < xkb) {
            double sinq = sin(q); y = sinq / rb; z = -rb * sinq; cosq = cos(q);
        } else if (wvno == xkb) {
            cosq = 1.0; y = (double)M->d[m]; z = 0.0;
        } else {
            double fac = 0.0;
            if (q < 16) fac = exp(-2.0 * q);
            cosq = (1.0 + fac) * 0.5;
            double sinq = (1.0 - fac) * 0.5;
            y = sinq / rb; z = rb * sinq;
        }
        double e10 = e1 * cosq + e2 * xmu * z;
        double e20 = e1 * y / xmu + e2 * cosq;
        double xnor = fabs(e10), ynor = fabs(e20);
        if (ynor > xnor) xnor = ynor;
        if (xnor < 1.0e-40) xnor = 1.0;
        e1 = e10 / xnor; e2 = e20 / xnor;
    }
    return e1;
}

/* surfdisp96.f:791-891  dltar4 (no water layer: llw = 1) */
static double sd_dltar4(double wvno, double omga, sd_model *M)
{
    if (M->ifunc == 1) return sd_dltar1(wvno, omga, M);
    double e[5], ee[5], ca[5][5];
    int mmax = M->n;
    M->nsec++;
    double omega = omga;
    if (omega < 1.0e-4) omega = 1.0e-4;
    double wvno2 = wvno * wvno;
    double xka = omega / (double)M->a[mmax - 1];
    double xkb = omega / (double)M->b[mmax - 1];
    double wvnop = wvno + xka, wvnom = fabs(wvno - xka);
    double ra = sqrt(wvnop * wvnom);
    wvnop = wvno + xkb; wvnom = fabs(wvno - xkb);
    double rb = sqrt(wvnop * wvnom);
    double t = (double)M->b[mmax - 1] / omega;
    double gammk = 2.0 * t * t;
    double gam = gammk * wvno2;
    double gamm1 = gam - 1.0;
    double rho1 = (double)M->rho[mmax - 1];
    e[0] = rho1 * rho1 * (gamm1 * gamm1 - gam * gammk * ra * rb);
    e[1] = -rho1 * ra;
    e[2] = rho1 * (gamm1 - gammk * ra * rb);
    e[3] = rho1 * rb;
    e[4] = wvno2 - ra * rb;
    for (int m = mmax - 2; m >= 0; m--) {
        xka = omega / (double)M->a[m];
        xkb = omega / (double)M->b[m];
        t = (double)M->b[m] / omega;
        gammk = 2.0 * t * t;
        gam = gammk * wvno2;
        wvnop = wvno + xka; wvnom = fabs(wvno - xka);
        ra = sqrt(wvnop * wvnom);
        wvnop = wvno + xkb; wvnom = fabs(wvno - xkb);
        rb = sqrt(wvnop * wvnom);
        double dpth = (double)M->d[m];
        rho1 = (double)M->rho[m];
        double p = ra * dpth, q = rb * dpth;
        double w, cosp, exa;
        sd_prod v;
        sd_var(p, q, ra, rb, wvno, xka, xkb, dpth, &w, &cosp, &exa, &v);
        sd_dnka(ca, wvno2, gam, gammk, rho1, &v);
        for (int i = 0; i < 5; i++) {
            double cr = 0.0;
            for (int j = 0; j < 5; j++) cr = cr + e[j] * ca[j][i];
            ee[i] = cr;
        }
        sd_normc(ee, &exa);
        for (int i = 0; i < 5; i++) e[i] = ee[i];
    }
    return e[0];
}

/* surfdisp96.f:375-396  gtsolh -- all single precision */
static float sd_gtsolh(float a, float b)
{
    float c = 0.95f * b;
    for (int i = 0; i < 5; i++) {
        float gamma = b / a;
        float kappa = c / b;
        float k2 = kappa * kappa;
        float gk2 = (gamma * kappa) * (gamma * kappa);
        float fac1 = sqrtf(1.0f - gk2);
        float fac2 = sqrtf(1.0f - k2);
        float fr = (2.0f - k2) * (2.0f - k2) - 4.0f * fac1 * fac2;
        float frp = -4.0f * (2.0f - k2) * kappa
                    + 4.0f * fac2 * gamma * gamma * kappa / fac1
                    + 4.0f * fac1 * kappa / fac2;
        frp = frp / b;
        c = c - fr / frp;
    }
    return c;
}

/* surfdisp96.f:689-701  half */
static void sd_half(double c1, double c2, double *c3, double *del3, double omega,
                    sd_model *M)
{
    *c3 = 0.5 * (c1 + c2);
    double wvno = omega / *c3;
    *del3 = sd_dltar4(wvno, omega, M);
}

/* surfdisp96.f:568-687  nevill */
static double sd_nevill(double t, double c1, double c2, double del1, double del2,
                        sd_model *M, double twopi)
{
    double x[21], y[21];
    double c3, del3;
    int m = 1;
    double omega = twopi / t;
    sd_half(c1, c2, &c3, &del3, omega, M);
    int nev = 1;
    int nctrl = 1;
    const double pct = (double)0.01f; /* default-real literal 0.01 (:637,639) */
    for (;;) {
        nctrl = nctrl + 1;
        if (nctrl >= 100) break;
        if (c3 < fmin(c1, c2) || c3 > fmax(c1, c2)) {
            nev = 0;
            sd_half(c1, c2, &c3, &del3, omega, M);
        }
        double s13 = del1 - del3;
        double s32 = del3 - del2;
        if (dsign1(del3) * dsign1(del1) < 0.0) {
            c2 = c3; del2 = del3;
        } else {
            c1 = c3; del1 = del3;
        }
        if (fabs(c1 - c2) <= 1.0e-6 * c1) break;
        if (dsign1(s13) != dsign1(s32)) nev = 0;
        double ss1 = fabs(del1), s1 = pct * ss1;
        double ss2 = fabs(del2), s2 = pct * ss2;
        if (s1 > ss2 || s2 > ss1 || nev == 0) {
            sd_half(c1, c2, &c3, &del3, omega, M);
            nev = 1; m = 1;
        } else {
            if (nev == 2) {
                x[m + 1] = c3; y[m + 1] = del3;
            } else {
                x[1] = c1; y[1] = del1; x[2] = c2; y[2] = del2; m = 1;
            }
            int bail = 0;
            for (int kk = 1; kk <= m; kk++) {
                int j = m - kk + 1;
                double denom = y[m + 1] - y[j];
                if (fabs(denom) < 1.0e-10 * fabs(y[m + 1])) { bail = 1; break; }
                x[j] = (-y[j] * x[j + 1] + y[m + 1] * x[j]) / denom;
            }
            if (!bail) {
                c3 = x[1];
                double wvno = omega / c3;
                del3 = sd_dltar4(wvno, omega, M);
                nev = 2;
                m = m + 1;
                if (m > 10) m = 10;
            } else {
                sd_half(c1, c2, &c3, &del3, omega, M);
                nev = 1; m = 1;
            }
        }
    }
    return c3;
}

/* surfdisp96.f:398-491  getsol; del1st is the routine's SAVEd variable (:423) */
static int sd_getsol(double t1, double *c1io, double clow, double dc, double cm,
                     float betmx, int ifirst, sd_model *M, double *del1st)
{
    const double twopi = 2.0 * 3.141592653589793;
    double c1 = *c1io, c2, del1, del2;
    double omega = twopi / t1;
    double wvno = omega / c1;
    int idir;
    del1 = sd_dltar4(wvno, omega, M);
    if (ifirst == 1) *del1st = del1;
    double plmn = dsign1(*del1st) * dsign1(del1);
    if (ifirst == 1) idir = +1;
    else if (plmn >= 0.0) idir = +1;
    else idir = -1;
    for (;;) {
        if (idir > 0) c2 = c1 + dc; else c2 = c1 - dc;
        if (c2 <= clow) { idir = +1; c1 = clow; continue; } /* del1 not redone (:463-467) */
        omega = twopi / t1;
        wvno = omega / c2;
        del2 = sd_dltar4(wvno, omega, M);
        if (dsign1(del1) != dsign1(del2)) break;
        c1 = c2; del1 = del2;
        if (c1 < cm) return -1;
        if (c1 >= ((double)betmx + dc)) return -1;
    }
    double cn = sd_nevill(t1, c1, c2, del1, del2, M, twopi);
    c1 = cn;
    if (c1 > (double)betmx) return -1;
    *c1io = c1;
    return 1;
}

/*
 * surfdisp96.f:54-368  surfdisp96 for iwave=2 (Rayleigh), igr=0, iflsph=0,
 * mode=1 (fundamental).  Returns ierr; cg[k] = float32-rounded phase velocity.
 */
/* surfdisp96.f:495-564  sphere: earth-flattening of the float32 work arrays (note ar = 6370 here,
 * 6371 in sregn96/slegn96; d(mmax) is forced to 1 first; powers in single precision) */
static void sd_sphere(int ifunc, int iflag, float *d, float *a, float *b, float *rho, float *rtp,
                      float *dtp, float *btp, int mmax, float *dhalf)
{
    double ar = 6370.0, dr = 0.0, r0 = ar, r1, z0, z1, tmp;
    d[mmax - 1] = 1.0f;
    if (iflag == 0) {
        for (int i = 0; i < mmax; i++) { dtp[i] = d[i]; rtp[i] = rho[i]; }
        for (int i = 0; i < mmax; i++) {
            dr = dr + (double)d[i];
            r1 = ar - dr;
            z0 = ar * log(ar / r0);
            z1 = ar * log(ar / r1);
            d[i] = (float)(z1 - z0);
            tmp = (ar + ar) / (r0 + r1);
            a[i] = (float)((double)a[i] * tmp);
            b[i] = (float)((double)b[i] * tmp);
            btp[i] = (float)tmp;
            r0 = r1;
        }
        *dhalf = d[mmax - 1];
    } else {
        d[mmax - 1] = *dhalf;
        for (int i = 0; i < mmax; i++) {
            if (ifunc == 1) {       /* btp**(-5): integer power by squaring, then reciprocal */
                float x2 = btp[i] * btp[i], x4 = x2 * x2, x5 = btp[i] * x4;
                rho[i] = rtp[i] * (1.0f / x5);
            }
            else if (ifunc == 2) rho[i] = rtp[i] * powf(btp[i], -2.275f);
        }
    }
    d[mmax - 1] = 0.0f;
}

static int sd_surfdisp96(const float *thkm, const float *vpm, const float *vsm,
                         const float *rhom, int nlayer, int kmax, const double *t,
                         double *cg, long *nsec, int iwave, int iflsph, int mode)
{
    float *wk = (float *)malloc(sizeof(float) * 7 * (size_t)nlayer);
    float *wd = wk, *wa = wk + nlayer, *wb = wk + 2 * nlayer, *wr = wk + 3 * nlayer;
    float *rtp = wk + 4 * nlayer, *dtp = wk + 5 * nlayer, *btp = wk + 6 * nlayer, dhalf = 0.0f;
    for (int i = 0; i < nlayer; i++) { wb[i] = vsm[i]; wa[i] = vpm[i]; wd[i] = thkm[i]; wr[i] = rhom[i]; }
    if (iflsph == 1) sd_sphere(0, 0, wd, wa, wb, wr, rtp, dtp, btp, nlayer, &dhalf);
    thkm = wd; vpm = wa; vsm = wb; rhom = wr;
    sd_model M = { nlayer, thkm, vpm, vsm, rhom, 0, iwave };
    int ierr = 0;
    float betmx = -1.e20f, betmn = 1.e20f;
    int jmn = 0, jsol = 1;
    for (int i = 0; i < nlayer; i++) {
        if (vsm[i] > 0.01f && vsm[i] < betmn) { betmn = vsm[i]; jmn = i; jsol = 1; }
        else if (vsm[i] <= 0.01f && vpm[i] < betmn) { betmn = vpm[i]; jmn = i; jsol = 0; }
        if (vsm[i] > betmx) betmx = vsm[i];
    }
    if (iflsph == 1) sd_sphere(iwave, 1, wd, wa, wb, wr, rtp, dtp, btp, nlayer, &dhalf);
    float ddc = 0.005f, sone = 1.5f;
    double one = 1.0e-2;                 /* `one` of surfdisp96.f (:190 one=1.0d-2) */
    double onea = (double)sone;
    float cc1;
    if (jsol == 0) cc1 = betmn;
    else cc1 = sd_gtsolh(vpm[jmn], vsm[jmn]);
    cc1 = 0.95f * cc1;
    cc1 = 0.90f * cc1;
    double cc = (double)cc1;
    double dc = fabs((double)ddc);
    double c1 = cc, cm = cc, clow;
    double *c = (double *)calloc((size_t)kmax, sizeof(double));
    double del1st = 0.0;
    /* :227-316, 337-362  the mode loop `do 1800 iq=1,mode`: mode iq searches above mode iq-1, whose roots c(k) it finds
     * in the same array it overwrites; cg(k) is overwritten mode after mode, so the LAST mode's values remain.  A mode
     * that is not found from period k on zeroes cg(k:) and caps every later mode at that period (ift); only the
     * fundamental sets ierr. */
    int ift = 1 << 30;
    for (int iq = 1; iq <= mode; iq++) {
        int k, failed = 0;
        for (k = 0; k < kmax; k++) {
            if (k >= ift) { failed = 1; break; }
            double t1 = t[k];
            int ifirst;
            if (k == 0 && iq == 1) { c1 = cc; clow = cc; ifirst = 1; }
            else if (k == 0 && iq > 1) { c1 = c[0] + one * dc; clow = c1; ifirst = 1; }
            else if (k > 0 && iq > 1) { ifirst = 0; clow = c[k] + one * dc; c1 = c[k - 1]; if (c1 < clow) c1 = clow; }
            else { ifirst = 0; c1 = c[k - 1] - onea * dc; clow = cm; }
            int iret = sd_getsol(t1, &c1, clow, dc, cm, betmx, ifirst, &M, &del1st);
            if (iret == -1) { failed = 1; break; }
            c[k] = c1;
            float cc0 = (float)c[k];
            cg[k] = (double)cc0;
        }
        if (failed) {
            if (iq == 1) ierr = 1;
            ift = k;
            for (int i = k; i < kmax; i++) cg[i] = 0.0;
        }
    }
    free(c);
    free(wk);
    if (nsec) *nsec += M.nsec;
    return ierr;
}

/* surfdisp.cpp:16-49  _flat2sphere: wave 'L' or 'R', kind 'c' (phase) or 'g' (group) */
static double sd_flat2sphere(double t, double c, char wave, char kind)
{
    double ar = 6371.0, omega = 2.0 * 3.14159265358979323846 / t, tm;
    if (wave == 'L') tm = 1. + pow(1.5 * c / (ar * omega), 2);
    else tm = 1. + pow(0.5 * c / (ar * omega), 2);
    tm = sqrt(tm);
    return (kind == 'c') ? c / tm : c * tm;
}

/* number of modes the search runs through (`mode + 1` of surfdisp.cpp:91: 1 = fundamental only); set by the *_m entries */
static int g_nmode = 1;

/* surfdisp.cpp:62-109  _surfdisp for phase velocities: iwave 1 Love / 2 Rayleigh */
static int sd_surfdisp(const float *thk, const float *vp, const float *vs, const float *rho,
                       int nlayer, const double *t, double *cg, int kmax, int iwave, int sphere,
                       int keep_flat, long *nsec)
{
    int ierr = sd_surfdisp96(thk, vp, vs, rho, nlayer, kmax, t, cg, nsec, iwave, sphere, g_nmode);
    if (ierr != 0) {
        for (int i = 0; i < kmax; i++) {
            if (cg[i] == 0.0 || isnan(cg[i])) {
                ierr = sd_surfdisp96(thk, vp, vs, rho, nlayer, 1, &t[i], &cg[i], nsec, iwave, sphere, g_nmode);
                if (ierr != 0) return ierr;
            }
        }
    }
    if (sphere && !keep_flat)
        for (int i = 0; i < kmax; i++) cg[i] = sd_flat2sphere(t[i], cg[i], iwave == 1 ? 'L' : 'R', 'c');
    return ierr;
}

/* surfdisp.cpp:62-109  _surfdisp ("Rc", flat, keep_flat) incl. the per-period retry */
int orc_surfdisp_rc(const float *thk, const float *vp, const float *vs, const float *rho,
                    int nlayer, const double *t, double *cg, int kmax, long *nsec)
{
    return sd_surfdisp(thk, vp, vs, rho, nlayer, t, cg, kmax, 2, 0, 1, nsec);
}

/* general phase-velocity entry: wave 1 Love / 2 Rayleigh, sphere 0/1, keep_flat 0/1 */
int orc_surfdisp(const float *thk, const float *vp, const float *vs, const float *rho, int nlayer,
                 const double *t, double *cg, int kmax, int iwave, int sphere, int keep_flat, long *nsec)
{
    return sd_surfdisp(thk, vp, vs, rho, nlayer, t, cg, kmax, iwave, sphere, keep_flat, nsec);
}

/* ================================================================== */
/*                      sregn96.f90 restatement                        */
/* ================================================================== */

typedef struct {
    int mmax;
    double vtp[ORC_MAXL], dtp[ORC_MAXL], rtp[ORC_MAXL];   /* earth-flattening factors (bldsph) */
    double zd[ORC_MAXL], zrho[ORC_MAXL], za[ORC_MAXL], zb[ORC_MAXL];
    double xmu[ORC_MAXL], xlam[ORC_MAXL];
    double ur[ORC_MAXL], uz[ORC_MAXL], tz[ORC_MAXL], tr[ORC_MAXL];
    double dcda[ORC_MAXL], dcdb[ORC_MAXL], dcdr[ORC_MAXL], dcdh[ORC_MAXL];
    double exe[ORC_MAXL], exa[ORC_MAXL], cd[ORC_MAXL][5], vv[ORC_MAXL][4];
    double sumi0, sumi1, sumi2, sumi3, flagr, are, ugr;
    zc ra, rb, e[4][4], einv[4][4];
} sr_state;

/* sregn96.f90:1405-1434  normc (n-vector) */
static void sr_normc(double *ee, double *ex, int nmat)
{
    double t1 = 0.0;
    for (int i = 0; i < nmat; i++)
        if (fabs(ee[i]) > t1) t1 = fabs(ee[i]);
    if (t1 < 1.0e-40) t1 = 1.0;
    for (int i = 0; i < nmat; i++) ee[i] = ee[i] / t1;
    *ex = log(t1);
}

/* sregn96.f90:831-915  varsv (elastic layer branch; iwat = 0) */
static void sr_varsv(zc p, zc q, zc rp, zc rsv, double *cosp, double *cosq,
                     double *rsinp, double *rsinq, double *sinpr, double *sinqr,
                     double *pex, double *svex, double zd)
{
    double pr = creal(p), pi = cimag(p), qr = creal(q), qi = cimag(q);
    double pfac, svfac;
    *pex = pr;
    *svex = qr;
    zc epp = (cos(pi) + sin(pi) * I) / 2.0;
    zc epm = conj(epp);
    zc eqp = (cos(qi) + sin(qi) * I) / 2.0;
    zc eqm = conj(eqp);
    if (pr < 30.) pfac = exp(-2. * pr); else pfac = 0.0;
    *cosp = creal(epp + pfac * epm);
    zc sinp = epp - pfac * epm;
    *rsinp = creal(rp * sinp);
    if (fabs(pr) < (double)1.0e-5f && cabs(rp) < (double)1.0e-5f) *sinpr = zd;
    else *sinpr = creal(sinp / rp);
    if (qr < 30.) svfac = exp(-2. * qr); else svfac = 0.0;
    *cosq = creal(eqp + svfac * eqm);
    zc sinq = eqp - svfac * eqm;
    *rsinq = creal(rsv * sinq);
    if (fabs(qr) < (double)1.0e-5f && cabs(rsv) < (double)1.0e-5f) *sinqr = zd;
    else *sinqr = creal(sinq / rsv);
}

/* sregn96.f90:494-650  dnka (hspec96-convention 5x5 compound matrix, elastic) */
static void sr_dnka(double ca[5][5], double cosp, double rsinp, double sinpr,
                    double cossv, double rsinsv, double sinsvr, float rho, float b,
                    double exa, double wvno, double wvno2, double om2)
{
    double a0;
    if (exa < 60.0) a0 = exp(-exa); else a0 = 0.0;
    double cpcq = cosp * cossv, cpy = cosp * sinsvr, cpz = cosp * rsinsv;
    double cqw = cossv * sinpr, cqx = cossv * rsinp;
    double xy = rsinp * sinsvr, xz = rsinp * rsinsv, wy = sinpr * sinsvr, wz = sinpr * rsinsv;
    float rho2f = rho * rho;                 /* real(c_float) rho2 (:508,596) */
    double rho2 = (double)rho2f, rhod = (double)rho;
    float twobb = (2.0f * b) * b;            /* 2.0*b*b evaluated in single (:597) */
    double gam = (double)twobb * wvno2 / om2;
    double gam2 = gam * gam, gamm1 = gam - 1., gamm2 = gamm1 * gamm1;
    double cqww2 = cqw * wvno2, cqxw2 = cqx / wvno2, gg1 = gam * gamm1;
    double a0c = 2.0 * (a0 - cpcq);
    double xz2 = xz / wvno2, gxz2 = gam * xz2, g2xz2 = gam2 * xz2;
    double a0cgg1 = a0c * (gam + gamm1);
    double wy2 = wy * wvno2, g2wy2 = gamm2 * wy2, g1wy2 = gamm1 * wy2;
    double temp = a0c * gg1 + g2xz2 + g2wy2;
    ca[2][2] = a0 + temp + temp;
    ca[0][0] = cpcq - temp;
    ca[0][1] = (-cqx + wvno2 * cpy) / (rhod * om2);
    temp = 0.5 * a0cgg1 + gxz2 + g1wy2;
    ca[0][2] = wvno * temp / (rhod * om2);
    ca[0][3] = (-cqww2 + cpz) / (rhod * om2);
    temp = wvno2 * (a0c + wy2) + xz;
    ca[0][4] = -temp / (rho2 * om2 * om2);
    ca[1][0] = (-gamm2 * cqw + gam2 * cpz / wvno2) * rhod * om2;
    ca[1][1] = cpcq;
    ca[1][2] = (gamm1 * cqww2 - gam * cpz) / wvno;
    ca[1][3] = -wz;
    ca[1][4] = ca[0][3];
    temp = 0.5 * a0cgg1 * gg1 + gam2 * gxz2 + gamm2 * g1wy2;
    ca[2][0] = -2.0 * temp * rhod * om2 / wvno;
    ca[2][1] = -wvno * (gam * cqxw2 - gamm1 * cpy) * 2.0;
    ca[2][3] = -2.0 * ca[1][2];
    ca[2][4] = -2.0 * ca[0][2];
    ca[3][0] = (-gam2 * cqxw2 + gamm2 * cpy) * rhod * om2;
    ca[3][1] = -xy;
    ca[3][2] = -ca[2][1] / 2.0;
    ca[3][3] = ca[1][1];
    ca[3][4] = ca[0][1];
    temp = gamm2 * (a0c * gam2 + g2wy2) + gam2 * g2xz2;
    ca[4][0] = -rho2 * om2 * om2 * temp / wvno2;
    ca[4][1] = ca[3][0];
    ca[4][2] = -ca[2][0] / 2.0;
    ca[4][3] = ca[1][0];
    ca[4][4] = ca[0][0];
}

/* sregn96.f90:917-991  hska (elastic layer) */
static void sr_hska(double AA[4][4], double cosp, double rsinp, double sinpr,
                    double tcossv, double trsinsv, double tsinsvr, float rho, float b,
                    double pex, double svex, double wvno, double wvno2, double om2)
{
    double dfac;
    if ((pex - svex) > 70.0) dfac = 0.0; else dfac = exp(svex - pex);
    double cossv = dfac * tcossv, rsinsv = dfac * trsinsv, sinsvr = dfac * tsinsvr;
    float twobb = (2.0f * b) * b;
    double gam = (double)twobb * wvno2 / om2;
    double gamm1 = gam - 1.0;
    double rhod = (double)rho;
    AA[0][0] = cossv + gam * (cosp - cossv);
    AA[0][1] = -wvno * gamm1 * sinpr + gam * rsinsv / wvno;
    AA[0][2] = -wvno * (cosp - cossv) / (rhod * om2);
    AA[0][3] = (wvno2 * sinpr - rsinsv) / (rhod * om2);
    AA[1][0] = gam * rsinp / wvno - wvno * gamm1 * sinsvr;
    AA[1][1] = cosp - gam * (cosp - cossv);
    AA[1][2] = (-rsinp + wvno2 * sinsvr) / (rhod * om2);
    AA[1][3] = -AA[0][2];
    AA[2][0] = rhod * om2 * gam * gamm1 * (cosp - cossv) / wvno;
    AA[2][1] = rhod * om2 * (-gamm1 * gamm1 * sinpr + gam * gam * rsinsv / wvno2);
    AA[2][2] = AA[1][1];
    AA[2][3] = -AA[0][1];
    AA[3][0] = rhod * om2 * (gam * gam * rsinp / wvno2 - gamm1 * gamm1 * sinsvr);
    AA[3][1] = -AA[2][0];
    AA[3][2] = -AA[1][0];
    AA[3][3] = AA[0][0];
}

/* sregn96.f90:652-829  evalg, jbdry = 0, elastic layer m (0-based):
 * fills S->e, S->einv, S->ra, S->rb and the half-space compound vector gbr[5] */
static void sr_evalg(sr_state *S, int m, zc gbr[5], double wvno, double om,
                     double om2, double wvno2)
{
    double xka = om / S->za[m];
    double xkb = (S->zb[m] > 0.0) ? om / S->zb[m] : 0.0;
    zc ra = csqrt(wvno2 - xka * xka + 0.0 * I);
    zc rb = csqrt(wvno2 - xkb * xkb + 0.0 * I);
    S->ra = ra; S->rb = rb;
    double gam = S->zb[m] * wvno / om;
    gam = 2.0 * (gam * gam);
    double gamm1 = gam - 1.0;
    double rho = S->zrho[m];
    zc (*E)[4] = S->e, (*EI)[4] = S->einv;
    E[0][0] = wvno; E[0][1] = rb; E[0][2] = wvno; E[0][3] = -rb;
    E[1][0] = ra; E[1][1] = wvno; E[1][2] = -ra; E[1][3] = wvno;
    E[2][0] = rho * om2 * gamm1; E[2][1] = rho * om2 * gam * rb / wvno;
    E[2][2] = rho * om2 * gamm1; E[2][3] = -rho * om2 * gam * rb / wvno;
    E[3][0] = rho * om2 * gam * ra / wvno; E[3][1] = rho * om2 * gamm1;
    E[3][2] = -rho * om2 * gam * ra / wvno; E[3][3] = rho * om2 * gamm1;
    EI[0][0] = 0.5 * gam / wvno; EI[0][1] = -0.5 * gamm1 / ra;
    EI[0][2] = -0.5 / (rho * om2); EI[0][3] = 0.5 * wvno / (rho * om2 * ra);
    EI[1][0] = -0.5 * gamm1 / rb; EI[1][1] = 0.5 * gam / wvno;
    EI[1][2] = 0.5 * wvno / (rho * om2 * rb); EI[1][3] = -0.5 / (rho * om2);
    EI[2][0] = 0.5 * gam / wvno; EI[2][1] = 0.5 * gamm1 / ra;
    EI[2][2] = -0.5 / (rho * om2); EI[2][3] = -0.5 * wvno / (rho * om2 * ra);
    EI[3][0] = 0.5 * gamm1 / rb; EI[3][1] = 0.5 * gam / wvno;
    EI[3][2] = -0.5 * wvno / (rho * om2 * rb); EI[3][3] = -0.5 / (rho * om2);
    if (gbr) {
        zc den = -rho * rho * om2 * om2 * wvno2 * ra * rb;
        gbr[0] = (rho * rho) * om2 * om2 * (-gam * gam * ra * rb + wvno2 * gamm1 * gamm1);
        gbr[1] = -rho * (wvno2 * ra) * om2;
        gbr[2] = -rho * (-gam * ra * rb + wvno2 * gamm1) * om2 * wvno;
        gbr[3] = rho * (wvno2 * rb) * om2;
        gbr[4] = wvno2 * (wvno2 - ra * rb);
        for (int i = 0; i < 5; i++) gbr[i] = 0.25 * gbr[i] / den;
    }
}

/* sregn96.f90:404-492  up */
static void sr_up(sr_state *S, double omega, double wvno, double *fr)
{
    int mmax = S->mmax;
    double wvno2 = wvno * wvno, om2 = omega * omega;
    zc gbr[5];
    double ca[5][5], ee[5];
    sr_evalg(S, mmax - 1, gbr, wvno, omega, om2, wvno2);
    for (int i = 0; i < 5; i++) S->cd[mmax - 1][i] = creal(gbr[i]);
    S->exe[mmax - 1] = 0.0;
    double exsum = 0.0;
    for (int m = mmax - 2; m >= 0; m--) {
        double xka = omega / S->za[m];
        double xkb = (S->zb[m] > 0.0) ? omega / S->zb[m] : 0.0;
        zc rp = csqrt(wvno2 - xka * xka + 0.0 * I);
        zc rsv = csqrt(wvno2 - xkb * xkb + 0.0 * I);
        zc p = rp * S->zd[m], q = rsv * S->zd[m];
        double cosp, cossv, rsinp, rsinsv, sinpr, sinsvr, pex, svex;
        sr_varsv(p, q, rp, rsv, &cosp, &cossv, &rsinp, &rsinsv, &sinpr, &sinsvr,
                 &pex, &svex, S->zd[m]);
        sr_dnka(ca, cosp, rsinp, sinpr, cossv, rsinsv, sinsvr, (float)S->zrho[m],
                (float)S->zb[m], pex + svex, wvno, wvno2, om2);
        for (int i = 0; i < 5; i++) {
            double cr = 0.0;
            for (int j = 0; j < 5; j++) cr = cr + S->cd[m + 1][j] * ca[j][i];
            ee[i] = cr;
        }
        double exn = 0.0;
        sr_normc(ee, &exn, 5);
        exsum = exsum + pex + svex + exn;
        S->exe[m] = exsum;
        for (int i = 0; i < 5; i++) S->cd[m][i] = ee[i];
    }
    *fr = S->cd[0][0];
}

/* sregn96.f90:993-1063  down */
static void sr_down(sr_state *S, double omega, double wvno)
{
    int mmax = S->mmax;
    double om2 = omega * omega, wvno2 = wvno * wvno;
    double aa[4][4], aa0[4];
    S->vv[0][0] = 1.0; S->vv[0][1] = 0.0; S->vv[0][2] = 0.0; S->vv[0][3] = 0.0;
    S->exa[0] = 0.0;
    double exsum = 0.0;
    for (int m = 0; m < mmax - 1; m++) {
        double xka = omega / S->za[m];
        double xkb = (S->zb[m] > 0.0) ? omega / S->zb[m] : 0.0;
        zc rp = csqrt(wvno2 - xka * xka + 0.0 * I);
        zc rsv = csqrt(wvno2 - xkb * xkb + 0.0 * I);
        zc p = rp * S->zd[m], q = rsv * S->zd[m];
        double cosp, cossv, rsinp, rsinsv, sinpr, sinsvr, pex, svex;
        sr_varsv(p, q, rp, rsv, &cosp, &cossv, &rsinp, &rsinsv, &sinpr, &sinsvr,
                 &pex, &svex, S->zd[m]);
        sr_hska(aa, cosp, rsinp, sinpr, cossv, rsinsv, sinsvr, (float)S->zrho[m],
                (float)S->zb[m], pex, svex, wvno, wvno2, om2);
        for (int i = 0; i < 4; i++) {
            double cc = 0.0;
            for (int j = 0; j < 4; j++) cc = cc + aa[i][j] * S->vv[m][j];
            aa0[i] = cc;
        }
        double ex2 = 0.0;
        sr_normc(aa0, &ex2, 4);
        exsum = exsum + pex + ex2;
        S->exa[m + 1] = exsum;
        for (int i = 0; i < 4; i++) S->vv[m + 1][i] = aa0[i];
    }
}

/* sregn96.f90:196-402  svfunc (no fluid layers) */
static void sr_svfunc(sr_state *S, double omega, double wvno)
{
    double fr;
    sr_up(S, omega, wvno, &fr);
    sr_down(S, omega, wvno);
    double f1213 = -S->cd[0][1];
    S->ur[0] = S->cd[0][2] / S->cd[0][1];
    S->uz[0] = 1.0; S->tz[0] = 0.0; S->tr[0] = 0.0;
    for (int i = 1; i < S->mmax; i++) {
        double cd1 = S->cd[i][0], cd2 = S->cd[i][1], cd3 = S->cd[i][2];
        double cd4 = -S->cd[i][2], cd5 = S->cd[i][3], cd6 = S->cd[i][4];
        double tz1 = -S->vv[i][3], tz2 = -S->vv[i][2], tz3 = S->vv[i][1], tz4 = S->vv[i][0];
        double uu1 = tz2 * cd6 - tz3 * cd5 + tz4 * cd4;
        double uu2 = -tz1 * cd6 + tz3 * cd3 - tz4 * cd2;
        double uu3 = tz1 * cd5 - tz2 * cd3 + tz4 * cd1;
        double uu4 = -tz1 * cd4 + tz2 * cd2 - tz3 * cd1;
        double ext = S->exa[i] + S->exe[i] - S->exe[0];
        if (ext > -80.0 && ext < 80.0) {
            double fact = exp(ext);
            S->ur[i] = uu1 * fact / f1213; S->uz[i] = uu2 * fact / f1213;
            S->tz[i] = uu3 * fact / f1213; S->tr[i] = uu4 * fact / f1213;
        } else {
            S->ur[i] = 0.0; S->uz[i] = 0.0; S->tz[i] = 0.0; S->tr[i] = 0.0;
        }
    }
}

/* sregn96.f90:1325-1403 */
static zc sr_ffunc(zc nub, double dm)
{
    if (cabs(nub) < 1.0e-08) return dm;
    zc argcd = nub * dm, exqq;
    if (creal(argcd) < 40.0) exqq = cexp(-2.0 * argcd); else exqq = 0.0;
    return (1.0 - exqq) / (2.0 * nub);
}
static zc sr_gfunc(zc nub, double dm)
{
    zc argcd = nub * dm;
    if (creal(argcd) < 75) return cexp(-argcd) * dm;
    return 0.0;
}
static zc sr_h1func(zc nua, zc nub, double dm)
{
    if (cabs(nub + nua) < 1.0e-08) return dm;
    zc argcd = (nua + nub) * dm, exqq;
    if (creal(argcd) < 40.0) exqq = cexp(-argcd); else exqq = 0.0;
    return (1.0 - exqq) / (nub + nua);
}
static zc sr_h2func(zc nua, zc nub, double dm)
{
    if (cabs(nub - nua) < 1.0e-08) return dm;
    zc argcd = nua * dm, exqp, exqq;
    if (creal(argcd) < 40.0) exqp = cexp(-argcd); else exqp = 0.0;
    argcd = nub * dm;
    if (creal(argcd) < 40.0) exqq = cexp(-argcd); else exqq = 0.0;
    return (exqq - exqp) / (nua - nub);
}

/* sregn96.f90:1203-1323  intijr (solid layer; typelyr 0 internal, +1 lower half-space);
 * i, j are 1-based as in the reference call sites */
static double sr_intijr(sr_state *S, int i, int j, int m, int typelyr, double om,
                        double om2, double wvno, double wvno2)
{
    sr_evalg(S, m, NULL, wvno, om, om2, wvno2);
    zc (*e)[4] = S->e, (*einv)[4] = S->einv;
    zc ra = S->ra, rb = S->rb, cint;
    i--; j--;
    if (typelyr == 0) {
        zc km1pd = einv[2][0] * S->ur[m] + einv[2][1] * S->uz[m] + einv[2][2] * S->tz[m] + einv[2][3] * S->tr[m];
        zc km1sd = einv[3][0] * S->ur[m] + einv[3][1] * S->uz[m] + einv[3][2] * S->tz[m] + einv[3][3] * S->tr[m];
        zc kmpu = einv[0][0] * S->ur[m + 1] + einv[0][1] * S->uz[m + 1] + einv[0][2] * S->tz[m + 1] + einv[0][3] * S->tr[m + 1];
        zc kmsu = einv[1][0] * S->ur[m + 1] + einv[1][1] * S->uz[m + 1] + einv[1][2] * S->tz[m + 1] + einv[1][3] * S->tr[m + 1];
        double dm = S->zd[m];
        zc FA = sr_ffunc(ra, dm), GA = sr_gfunc(ra, dm);
        zc FB = sr_ffunc(rb, dm), GB = sr_gfunc(rb, dm);
        zc H1 = sr_h1func(ra, rb, dm), H2 = sr_h2func(ra, rb, dm);
        cint = e[i][0] * e[j][0] * kmpu * kmpu * FA
             + e[i][2] * e[j][2] * km1pd * km1pd * FA
             + e[i][1] * e[j][1] * kmsu * kmsu * FB
             + e[i][3] * e[j][3] * km1sd * km1sd * FB
             + H1 * ((e[i][0] * e[j][1] + e[i][1] * e[j][0]) * kmpu * kmsu +
                     (e[i][2] * e[j][3] + e[i][3] * e[j][2]) * km1pd * km1sd)
             + H2 * ((e[i][0] * e[j][3] + e[i][3] * e[j][0]) * kmpu * km1sd +
                     (e[i][1] * e[j][2] + e[i][2] * e[j][1]) * km1pd * kmsu)
             + GA * (e[i][0] * e[j][2] + e[i][2] * e[j][0]) * kmpu * km1pd
             + GB * (e[i][1] * e[j][3] + e[i][3] * e[j][1]) * kmsu * km1sd;
    } else {
        zc km1pd = einv[2][0] * S->ur[m] + einv[2][1] * S->uz[m] + einv[2][2] * S->tz[m] + einv[2][3] * S->tr[m];
        zc km1sd = einv[3][0] * S->ur[m] + einv[3][1] * S->uz[m] + einv[3][2] * S->tz[m] + einv[3][3] * S->tr[m];
        cint = e[i][2] * e[j][2] * km1pd * km1pd / (2.0 * ra)
             + (e[i][2] * e[j][3] + e[i][3] * e[j][2]) * km1pd * km1sd / (ra + rb)
             + e[i][3] * e[j][3] * km1sd * km1sd / (2.0 * rb);
    }
    return creal(cint);
}

/* sregn96.f90:1436-1535  getdcdh (all-solid model) */
static void sr_getdcdh(sr_state *S, double om2, double wvno, double wvno2, double fac)
{
    for (int m = 0; m < S->mmax; m++) {
        double tuz = S->uz[m], ttz = S->tz[m], ttr = S->tr[m], tur = S->ur[m];
        double gfac1, gfac2, gfac3, gfac4, gfac5, gfac6;
        if (m == 0) {
            double drho = S->zrho[0], dmu = S->xmu[0], dlm = S->xlam[0];
            double dl2mu = dlm + dmu + dmu;
            double xl2mp = S->xlam[m] + S->xmu[m] + S->xmu[m];
            double duzdzp = (ttz + wvno * S->xlam[m] * tur) / xl2mp;
            double durdzp = (ttr / S->xmu[m]) - wvno * tuz;
            double drur2 = tur * tur * drho, dlur2 = tur * tur * dl2mu;
            gfac1 = om2 * drho * tuz * tuz;
            gfac2 = om2 * drur2;
            gfac3 = -wvno2 * dmu * tuz * tuz;
            gfac4 = -wvno2 * dlur2;
            gfac5 = (xl2mp * duzdzp * duzdzp);
            gfac6 = (S->xmu[m] * durdzp * durdzp);
        } else {
            double drho = S->zrho[m] - S->zrho[m - 1];
            double dmu = S->xmu[m] - S->xmu[m - 1];
            double dlm = S->xlam[m] - S->xlam[m - 1];
            double dl2mu = dlm + dmu + dmu;
            double xl2mp = S->xlam[m] + S->xmu[m] + S->xmu[m];
            double xl2mm = S->xlam[m - 1] + S->xmu[m - 1] + S->xmu[m - 1];
            double duzdzp = (ttz + wvno * S->xlam[m] * tur) / xl2mp;
            double durdzp = (S->xmu[m] == 0.0) ? wvno * tuz : (ttr / S->xmu[m]) - wvno * tuz;
            double durdzm = (S->xmu[m - 1] == 0.0) ? wvno * tuz : (ttr / S->xmu[m - 1]) - wvno * tuz;
            double drur2 = tur * tur * drho, dlur2 = tur * tur * dl2mu;
            double duzdzm = (ttz + wvno * S->xlam[m - 1] * tur) / xl2mm;
            gfac1 = om2 * drho * tuz * tuz;
            gfac2 = om2 * drur2;
            gfac3 = -wvno2 * dmu * tuz * tuz;
            gfac4 = -wvno2 * dlur2;
            gfac5 = (xl2mp * duzdzp * duzdzp - xl2mm * duzdzm * duzdzm);
            gfac6 = (S->xmu[m] * durdzp * durdzp - S->xmu[m - 1] * durdzm * durdzm);
        }
        double dfac = fac * (gfac1 + gfac2 + gfac3 + gfac4 + gfac5 + gfac6);
        if (fabs(dfac) < 1.0e-38) dfac = 0.0;
        S->dcdh[m] = dfac;
    }
}

/* sregn96.f90:1065-1201  energy (+ getmat :1537-1589, elastic branch) */
static void sr_energy(sr_state *S, double om, double wvno)
{
    int mmax = S->mmax;
    S->sumi0 = S->sumi1 = S->sumi2 = S->sumi3 = 0.0;
    double c = om / wvno, om2 = om * om, wvno2 = wvno * wvno;
    for (int m = 0; m < mmax; m++) {
        double ah = S->za[m], av = S->za[m], bv = S->zb[m], rho = S->zrho[m], eta = 1.0;
        double TL = S->zrho[m] * S->zb[m] * S->zb[m], TN = TL;
        double TC = S->zrho[m] * S->za[m] * S->za[m], TA = TC;
        double TF = TA - 2. * TN;
        double a12 = -wvno, a14 = 1.0 / TL, a21 = wvno * TF / TC, a23 = 1.0 / TC;
        int typelyr = (m == mmax - 1) ? 1 : 0;
        double INT11 = sr_intijr(S, 1, 1, m, typelyr, om, om2, wvno, wvno2);
        double INT13 = sr_intijr(S, 1, 3, m, typelyr, om, om2, wvno, wvno2);
        double INT22 = sr_intijr(S, 2, 2, m, typelyr, om, om2, wvno, wvno2);
        double INT24 = sr_intijr(S, 2, 4, m, typelyr, om, om2, wvno, wvno2);
        double INT33 = sr_intijr(S, 3, 3, m, typelyr, om, om2, wvno, wvno2);
        double INT44 = sr_intijr(S, 4, 4, m, typelyr, om, om2, wvno, wvno2);
        double URUR = INT11, UZUZ = INT22;
        double DURDUR = a12 * a12 * INT22 + 2. * a12 * a14 * INT24 + a14 * a14 * INT44;
        double DUZDUZ = a21 * a21 * INT11 + 2. * a21 * a23 * INT13 + a23 * a23 * INT33;
        double URDUZ = a21 * INT11 + a23 * INT13;
        double UZDUR = a12 * INT22 + a14 * INT24;
        S->sumi0 = S->sumi0 + rho * (URUR + UZUZ);
        S->sumi1 = S->sumi1 + TL * UZUZ + TA * URUR;
        S->sumi2 = S->sumi2 + TL * UZDUR - TF * URDUZ;
        S->sumi3 = S->sumi3 + TL * DURDUR + TC * DUZDUZ;
        double facah = rho * ah * (URUR - 2. * eta * URDUZ / wvno);
        double facav = rho * av * DUZDUZ / wvno2;
        double facbh = 0.0;
        double facbv = rho * bv * (UZUZ + 2. * UZDUR / wvno + DURDUR / wvno2 + 4. * eta * URDUZ / wvno);
        S->dcda[m] = facah + facav;
        S->dcdb[m] = facbv + facbh;
        double facr = -0.5 * c * c * (URUR + UZUZ);
        S->dcdr[m] = 0.5 * (av * facav + ah * facah + bv * facbv) / rho + facr;
    }
    S->flagr = om2 * S->sumi0 - wvno2 * S->sumi1 - 2.0 * wvno * S->sumi2 - S->sumi3;
    S->ugr = (wvno * S->sumi1 + S->sumi2) / (om * S->sumi0);
    S->are = wvno / (2.0 * om * S->ugr * S->sumi0);
    double fac = S->are * c / wvno2;
    for (int m = 0; m < mmax; m++) {
        S->dcda[m] = S->dcda[m] / (S->ugr * S->sumi0);
        S->dcdb[m] = S->dcdb[m] / (S->ugr * S->sumi0);
        S->dcdr[m] = S->dcdr[m] / (S->ugr * S->sumi0);
    }
    sr_getdcdh(S, om2, wvno, wvno2, fac);
}

/* sregn96.f90:133-187  bldsph (Rayleigh: density exponent -2.275, ar = 6371) */
static void sr_bldsph(sr_state *S)
{
    double ar = 6371.0, dr = 0.0, r0 = ar, r1, z0, z1, tmp;
    int mmax = S->mmax;
    for (int i = 0; i < mmax; i++) {
        if (i == mmax - 1) dr = dr + 1.0; else dr = dr + S->zd[i];
        r1 = ar - dr;
        z0 = ar * log(ar / r0);
        z1 = ar * log(ar / r1);
        tmp = (2.0 * ar) / (r0 + r1);
        S->vtp[i] = tmp;
        S->rtp[i] = pow(tmp, (double)-2.275f);    /* default-real exponent literal */
        S->dtp[i] = ar / r0;
        S->za[i] = S->za[i] * tmp; S->zb[i] = S->zb[i] * tmp;
        S->zrho[i] = S->zrho[i] * S->rtp[i];
        S->zd[i] = z1 - z0;
        r0 = r1;
    }
    S->zd[mmax - 1] = 0.0;
}

static void sr_load_model(sr_state *S, const float *thk, const float *vp, const float *vs,
                          const float *rhom, int nlayer, int iflsph)
{
    S->mmax = nlayer;
    for (int i = 0; i < nlayer; i++) {
        S->zb[i] = (double)vs[i]; S->za[i] = (double)vp[i];
        S->zrho[i] = (double)rhom[i]; S->zd[i] = (double)thk[i];
    }
    if (iflsph > 0) sr_bldsph(S);
    for (int i = 0; i < nlayer; i++) {
        S->xmu[i] = S->zrho[i] * (S->zb[i] * S->zb[i]);
        S->xlam[i] = S->zrho[i] * (S->za[i] * S->za[i]) - 2 * S->xmu[i];
    }
}

/* float32 pi: `pi = atan(1.0) * 4.0` in default real (sregn96.f90:1654,1773) */
static const double SR_PI = (double)3.14159274101257324f;

/* thickness kernels: interface partials -> layer-thickness partials (:1727-1731) */
static void sr_suffix_sum(double *dcdh, int mmax)
{
    for (int i = 0; i < mmax - 1; i++) {
        double sums = 0.0;
        for (int j = i + 1; j < mmax; j++) sums = sums + dcdh[j];
        dcdh[i] = sums;
    }
    dcdh[mmax - 1] = 0.0;
}

/* sregn96.f90:1637-1745  sregn96; *cp is replaced by the spherical phase velocity when iflsph > 0 */
void orc_sregn96s(const float *thk, const float *vp, const float *vs, const float *rhom,
                  int nlayer, double t, double *cp, double *cg, double *dc2da, double *dc2db,
                  double *dc2dh, double *dc2dr, int iflsph)
{
    sr_state *S = (sr_state *)calloc(1, sizeof(sr_state));
    sr_load_model(S, thk, vp, vs, rhom, nlayer, iflsph);
    double twopi = 2.0 * SR_PI;
    double omega = twopi / t;
    double c = *cp;
    double wvno = omega / c;
    sr_svfunc(S, omega, wvno);
    sr_energy(S, omega, wvno);
    if (fabs(S->ugr) < 1.0e-36) S->ugr = 0.0;
    double csph = c, usph = S->ugr;
    if (iflsph > 0) {           /* sprayl, sregn96.f90:1591-1635 */
        double ar = 6371.0, tm = sqrt(1. + (c / (2. * ar * omega)) * (c / (2. * ar * omega)));
        for (int i = 0; i < nlayer; i++) {
            S->dcda[i] = S->dcda[i] * S->vtp[i] / (tm * tm * tm);
            S->dcdb[i] = S->dcdb[i] * S->vtp[i] / (tm * tm * tm);
            S->dcdh[i] = S->dcdh[i] * S->dtp[i] / (tm * tm * tm);
            S->dcdr[i] = S->dcdr[i] * S->rtp[i] / (tm * tm * tm);
        }
        usph = S->ugr * tm; csph = c / tm;
    }
    sr_suffix_sum(S->dcdh, nlayer);
    for (int i = 0; i < nlayer; i++) {
        dc2da[i] = S->dcda[i]; dc2db[i] = S->dcdb[i];
        dc2dr[i] = S->dcdr[i]; dc2dh[i] = S->dcdh[i];
    }
    *cp = csph; *cg = usph;
    free(S);
}

void orc_sregn96(const float *thk, const float *vp, const float *vs, const float *rhom,
                 int nlayer, double t, double cp, double *cg, double *ur, double *uz,
                 double *tr, double *tz, double *dc2da, double *dc2db, double *dc2dh,
                 double *dc2dr)
{
    (void)ur; (void)uz; (void)tr; (void)tz;
    orc_sregn96s(thk, vp, vs, rhom, nlayer, t, &cp, cg, dc2da, dc2db, dc2dh, dc2dr, 0);
}

/* sregn96.f90:1747-1888  sregnpu.  The first term of du/dm uses the module arrays dcda.. which at
 * that point hold the t2 pass (quirk, :1841-1844).  *cp, *cg become spherical when iflsph > 0. */
void orc_sregnpus(const float *thk, const float *vp, const float *vs, const float *rhom,
                  int nlayer, double t, double *cp, double *cg, double t1, double cp1,
                  double t2, double cp2, double *dc2da, double *dc2db, double *dc2dh,
                  double *dc2dr, double *du2da, double *du2db, double *du2dh, double *du2dr, int iflsph)
{
    sr_state *S = (sr_state *)calloc(1, sizeof(sr_state));
    double *w = (double *)calloc((size_t)8 * nlayer, sizeof(double));
    double *a1 = w, *b1 = w + nlayer, *r1 = w + 2 * nlayer, *h1 = w + 3 * nlayer;
    double *a2 = w + 4 * nlayer, *b2 = w + 5 * nlayer, *r2 = w + 6 * nlayer, *h2 = w + 7 * nlayer;
    sr_load_model(S, thk, vp, vs, rhom, nlayer, iflsph);
    double twopi = 2.0 * SR_PI;
    double omega = twopi / t, wvno = omega / *cp;
    sr_svfunc(S, omega, wvno); sr_energy(S, omega, wvno);
    *cg = S->ugr;
    for (int i = 0; i < nlayer; i++) {
        dc2da[i] = S->dcda[i]; dc2db[i] = S->dcdb[i]; dc2dr[i] = S->dcdr[i]; dc2dh[i] = S->dcdh[i];
    }
    omega = twopi / t1; wvno = omega / cp1;
    sr_svfunc(S, omega, wvno); sr_energy(S, omega, wvno);
    for (int i = 0; i < nlayer; i++) {
        a1[i] = S->dcda[i]; b1[i] = S->dcdb[i]; r1[i] = S->dcdr[i]; h1[i] = S->dcdh[i];
    }
    omega = twopi / t2; wvno = omega / cp2;
    sr_svfunc(S, omega, wvno); sr_energy(S, omega, wvno);
    for (int i = 0; i < nlayer; i++) {
        a2[i] = S->dcda[i]; b2[i] = S->dcdb[i]; r2[i] = S->dcdr[i]; h2[i] = S->dcdh[i];
    }
    double uc1 = *cg / *cp;
    for (int i = 0; i < nlayer; i++) {
        du2da[i] = uc1 * (2.0 - uc1) * S->dcda[i] - uc1 * uc1 * t * (a2[i] - a1[i]) / (t2 - t1);
        du2db[i] = uc1 * (2.0 - uc1) * S->dcdb[i] - uc1 * uc1 * t * (b2[i] - b1[i]) / (t2 - t1);
        du2dr[i] = uc1 * (2.0 - uc1) * S->dcdr[i] - uc1 * uc1 * t * (r2[i] - r1[i]) / (t2 - t1);
        du2dh[i] = uc1 * (2.0 - uc1) * S->dcdh[i] - uc1 * uc1 * t * (h2[i] - h1[i]) / (t2 - t1);
    }
    if (iflsph > 0) {           /* sregn96.f90:1848-1868 */
        double ar = 6371.0;
        omega = twopi / t;
        double tm = sqrt(1. + (*cp / (2. * ar * omega)) * (*cp / (2. * ar * omega)));
        double tm1 = (0.5 / (ar * omega)) * (0.5 / (ar * omega)) / tm;
        for (int i = 0; i < nlayer; i++) {
            du2da[i] = (tm * du2da[i] + *cg * *cp * dc2da[i] * tm1) * S->vtp[i];
            du2db[i] = (tm * du2db[i] + *cg * *cp * dc2db[i] * tm1) * S->vtp[i];
            du2dr[i] = (tm * du2dr[i] + *cg * *cp * dc2dr[i] * tm1) * S->rtp[i];
            du2dh[i] = (tm * du2dh[i] + *cg * *cp * dc2dh[i] * tm1) * S->dtp[i];
            dc2da[i] = dc2da[i] / (tm * tm * tm) * S->vtp[i];
            dc2db[i] = dc2db[i] / (tm * tm * tm) * S->vtp[i];
            dc2dr[i] = dc2dr[i] / (tm * tm * tm) * S->rtp[i];
            dc2dh[i] = dc2dh[i] / (tm * tm * tm) * S->dtp[i];
        }
        *cp = *cp / tm; *cg = *cg * tm;
    }
    sr_suffix_sum(dc2dh, nlayer);
    sr_suffix_sum(du2dh, nlayer);
    free(w); free(S);
}

void orc_sregnpu(const float *thk, const float *vp, const float *vs, const float *rhom,
                 int nlayer, double t, double cp, double *cg, double t1, double cp1,
                 double t2, double cp2, double *dc2da, double *dc2db, double *dc2dh,
                 double *dc2dr, double *du2da, double *du2db, double *du2dh, double *du2dr)
{
    orc_sregnpus(thk, vp, vs, rhom, nlayer, t, &cp, cg, t1, cp1, t2, cp2, dc2da, dc2db, dc2dh, dc2dr,
                 du2da, du2db, du2dh, du2dr, 0);
}

/* ================================================================== */
/*                      slegn96.f90 restatement (Love)                 */
/* ================================================================== */
typedef struct {
    int mmax;
    double zd[ORC_MAXL], zb[ORC_MAXL], zrho[ORC_MAXL], xmu[ORC_MAXL];
    double vtp[ORC_MAXL], dtp[ORC_MAXL], rtp[ORC_MAXL];
    double uu[ORC_MAXL], tt[ORC_MAXL], exl[ORC_MAXL];
    double dcdb[ORC_MAXL], dcdh[ORC_MAXL], dcdr[ORC_MAXL];
    double sumi0, sumi1, sumi2, ugr;
} sl_state;

/* slegn96.f90:107-167  bldsph (Love: density exponent -5) */
static void sl_bldsph(sl_state *S)
{
    double ar = 6371.0, dr = 0.0, r0 = ar, r1, z0, z1, tmp;
    int mmax = S->mmax;
    S->zd[mmax - 1] = 1.0;
    for (int i = 0; i < mmax; i++) {
        dr = dr + S->zd[i];
        r1 = ar - dr;
        z0 = ar * log(ar / r0);
        z1 = ar * log(ar / r1);
        S->dtp[i] = ar / r0;
        tmp = (2.0 * ar) / (r0 + r1);
        S->vtp[i] = tmp;
        { double t2 = tmp * tmp, t4 = t2 * t2; S->rtp[i] = 1.0 / (tmp * t4); }   /* tmp**(-5) */
        S->zb[i] = S->zb[i] * tmp;
        S->zrho[i] = S->zrho[i] * S->rtp[i];
        S->zd[i] = z1 - z0;
        r0 = r1;
    }
    S->zd[mmax - 1] = 0.0;
}

typedef struct { double cosq, yl, zl, mu, rb, xkb, eexl; } sl_var;

/* slegn96.f90:248-328  varl */
static void sl_varl(const sl_state *S, int m, double omega, double wvno, double dpth, sl_var *v)
{
    v->xkb = omega / S->zb[m];
    double wvnop = wvno + v->xkb, wvnom = fabs(wvno - v->xkb);
    v->rb = sqrt(wvnop * wvnom);
    double q = v->rb * dpth;
    v->mu = S->zrho[m] * S->zb[m] * S->zb[m];
    v->eexl = 0.0;
    if (wvno < v->xkb) {
        double sinq = sin(q); v->yl = sinq / v->rb; v->zl = -v->rb * sinq; v->cosq = cos(q);
    } else if (wvno == v->xkb) {
        v->cosq = 1.0; v->yl = dpth; v->zl = 0.0;
    } else {
        v->eexl = q;
        double fac = 0.0;
        if (q < 18.0) fac = exp(-2.0 * q);
        v->cosq = (1.0 + fac) * 0.5;
        double sinq = (1.0 - fac) * 0.5;
        v->yl = sinq / v->rb; v->zl = v->rb * sinq;
    }
}

/* slegn96.f90:372-445  up  +  :179-246 shfunc (all-solid model) */
static void sl_shfunc(sl_state *S, double omega, double wvno)
{
    int mmax = S->mmax;
    sl_var v;
    if (S->zb[mmax - 1] > 0.01) {
        sl_varl(S, mmax - 1, omega, wvno, 0.0, &v);
        S->uu[mmax - 1] = 1.0; S->tt[mmax - 1] = -S->xmu[mmax - 1] * v.rb;
    } else { S->uu[mmax - 1] = 1.0; S->tt[mmax - 1] = 0.0; }
    S->exl[mmax - 1] = 0.0;
    for (int k = mmax - 2; k >= 0; k--) {
        sl_varl(S, k, omega, wvno, S->zd[k], &v);
        double a11 = v.cosq, a22 = v.cosq, a12 = -(v.yl / v.mu), a21 = -(v.zl * v.mu);
        double amp0 = a11 * S->uu[k + 1] + a12 * S->tt[k + 1];
        double str0 = a21 * S->uu[k + 1] + a22 * S->tt[k + 1];
        double rr = fabs(amp0), ss = fabs(str0);
        if (ss > rr) rr = ss;
        if (rr < 1.0e-30) rr = 1.0;
        S->exl[k] = log(rr) + v.eexl;
        S->uu[k] = amp0 / rr; S->tt[k] = str0 / rr;
    }
    double ext = 0.0, umax = S->uu[0];
    S->tt[0] = 0.0;
    for (int k = 1; k < mmax; k++) {
        ext = ext + S->exl[k - 1];
        double fact = 0.0;
        if (ext < 80.0) fact = 1. / exp(ext);
        S->uu[k] = S->uu[k] * fact; S->tt[k] = S->tt[k] * fact;
        if (fabs(S->uu[k]) > fabs(umax)) umax = S->uu[k];
    }
    if (S->uu[0] != 0.0) umax = S->uu[0];
    if (fabs(umax) > 0.0)
        for (int k = 0; k < mmax; k++) { S->uu[k] = S->uu[k] / umax; S->tt[k] = S->tt[k] / umax; }
}

/* slegn96.f90:447-629  energy */
static void sl_energy(sl_state *S, double omega, double wvno)
{
    int mmax = S->mmax;
    double c = omega / wvno, omega2 = omega * omega, wvno2 = wvno * wvno;
    S->sumi0 = S->sumi1 = S->sumi2 = 0.0;
    for (int k = 0; k < mmax; k++) {
        double TN = S->zrho[k] * S->zb[k] * S->zb[k], TL = TN, VSHH = S->zb[k], VSHV = S->zb[k];
        double drho = S->zrho[k], dpth = S->zd[k], dmu = S->xmu[k], upup, dupdup;
        sl_var v;
        sl_varl(S, k, omega, wvno, dpth, &v);
        double rb = v.rb;
        if (rb < 1.0e-10) rb = 1.0e-10;
        if (k == mmax - 1) {
            upup = (0.5 / rb) * S->uu[mmax - 1] * S->uu[mmax - 1];
            dupdup = (0.5 * rb) * S->uu[mmax - 1] * S->uu[mmax - 1];
        } else {
            zc nub = rb + 0.0 * I;
            if (wvno < v.xkb) nub = 0.0 + rb * I;
            zc xnub = dmu * nub;
            zc ei11 = 0.5 / wvno, ei12 = 0.5 / (wvno * xnub), ei21 = 0.5 / wvno, ei22 = -0.5 / (wvno * xnub);
            zc el11 = wvno, el12 = wvno;
            zc km1dn = ei21 * S->uu[k] + ei22 * S->tt[k];
            zc kmup = ei11 * S->uu[k + 1] + ei12 * S->tt[k + 1];
            zc f3 = nub * dpth, exqq = 0.0;
            if (creal(f3) < 40.0) exqq = cexp(-2.0 * f3);
            zc f = (1.0 - exqq) / (2.0 * nub);
            exqq = 0.0;
            if (creal(f3) < 75.0) exqq = cexp(-f3);
            zc g = dpth * exqq;
            zc f1 = f * (el11 * el11 * kmup * kmup + el12 * el12 * km1dn * km1dn);
            zc f2 = g * (el11 * el12 + el11 * el12) * kmup * km1dn;
            upup = creal(f1 + f2);
            dupdup = creal(nub * nub * (f1 - f2));
        }
        S->sumi0 += drho * upup; S->sumi1 += TN * upup; S->sumi2 += TL * dupdup;
        double DCDBH = c * drho * VSHH * upup, DCDBV = c * drho * VSHV * dupdup / wvno2;
        S->dcdb[k] = DCDBH + DCDBV;
        S->dcdr[k] = 0.5 * c * (-c * c * upup + VSHH * VSHH * upup + VSHV * VSHV * dupdup / wvno2);
    }
    for (int k = 0; k < mmax; k++) { S->dcdb[k] /= S->sumi1; S->dcdr[k] /= S->sumi1; }
    S->ugr = S->sumi1 / (c * S->sumi0);
    double ale = 0.5 / S->sumi1, fac = ale * c / wvno2;
    for (int k = 0; k < mmax; k++) {
        double drho, dmu, dvdz;
        if (k == 0) { drho = S->zrho[k]; dmu = S->xmu[k]; dvdz = 0.0; }
        else {
            drho = S->zrho[k] - S->zrho[k - 1]; dmu = S->xmu[k] - S->xmu[k - 1];
            dvdz = S->tt[k] * S->tt[k] * (1.0 / S->xmu[k] - 1.0 / S->xmu[k - 1]);
        }
        double dfac = fac * (S->uu[k] * S->uu[k] * (omega2 * drho - wvno2 * dmu) + dvdz);
        S->dcdh[k] = (fabs(dfac) < 1.0e-38) ? 0.0 : dfac;
    }
}

static void sl_load(sl_state *S, const float *thk, const float *vs, const float *rhom, int nlayer, int iflsph)
{
    S->mmax = nlayer;
    for (int i = 0; i < nlayer; i++) { S->zb[i] = vs[i]; S->zrho[i] = rhom[i]; S->zd[i] = thk[i]; }
    if (iflsph > 0) sl_bldsph(S);
    for (int i = 0; i < nlayer; i++) S->xmu[i] = S->zrho[i] * (S->zb[i] * S->zb[i]);
}

/* slegn96.f90:672-783  slegn96; pi is the default-real literal 3.1415926535898 -> float32 pi (:689) */
void orc_slegn96(const float *thk, const float *vs, const float *rhom, int nlayer, double t, double *cp,
                 double *cg, double *dc2db, double *dc2dh, double *dc2dr, int iflsph)
{
    sl_state *S = (sl_state *)calloc(1, sizeof(sl_state));
    sl_load(S, thk, vs, rhom, nlayer, iflsph);
    double omega = 2.0 * SR_PI / t, c = *cp, wvno = omega / c;
    sl_shfunc(S, omega, wvno);
    sl_energy(S, omega, wvno);
    double csph = c, usph = S->ugr;
    if (iflsph > 0) {           /* splove, slegn96.f90:631-670 */
        double a = 6371.0, tm = sqrt(1. + (3.0 * c / (2. * a * omega)) * (3.0 * c / (2. * a * omega)));
        for (int i = 0; i < nlayer; i++) {
            S->dcdb[i] = S->dcdb[i] * S->vtp[i] / (tm * tm * tm);
            S->dcdh[i] = S->dcdh[i] * S->dtp[i] / (tm * tm * tm);
            S->dcdr[i] = S->dcdr[i] * S->rtp[i] / (tm * tm * tm);
        }
        csph = c / tm; usph = S->ugr * tm;
    }
    sr_suffix_sum(S->dcdh, nlayer);
    for (int i = 0; i < nlayer; i++) { dc2db[i] = S->dcdb[i]; dc2dr[i] = S->dcdr[i]; dc2dh[i] = S->dcdh[i]; }
    *cp = csph; *cg = usph;
    free(S);
}

/* slegn96.f90:785-919  slegnpu */
void orc_slegnpu(const float *thk, const float *vs, const float *rhom, int nlayer, double t, double *cp,
                 double *cg, double t1, double cp1, double t2, double cp2, double *dc2db, double *dc2dh,
                 double *dc2dr, double *du2db, double *du2dh, double *du2dr, int iflsph)
{
    sl_state *S = (sl_state *)calloc(1, sizeof(sl_state));
    double *w = (double *)calloc((size_t)6 * nlayer, sizeof(double));
    double *b1 = w, *r1 = w + nlayer, *h1 = w + 2 * nlayer, *b2 = w + 3 * nlayer, *r2 = w + 4 * nlayer, *h2 = w + 5 * nlayer;
    sl_load(S, thk, vs, rhom, nlayer, iflsph);
    double twopi = 2.0 * SR_PI, omega = twopi / t, wvno = omega / *cp;
    sl_shfunc(S, omega, wvno); sl_energy(S, omega, wvno);
    *cg = S->ugr;
    for (int i = 0; i < nlayer; i++) { dc2db[i] = S->dcdb[i]; dc2dr[i] = S->dcdr[i]; dc2dh[i] = S->dcdh[i]; }
    omega = twopi / t1; wvno = omega / cp1;
    sl_shfunc(S, omega, wvno); sl_energy(S, omega, wvno);
    for (int i = 0; i < nlayer; i++) { b1[i] = S->dcdb[i]; r1[i] = S->dcdr[i]; h1[i] = S->dcdh[i]; }
    omega = twopi / t2; wvno = omega / cp2;
    sl_shfunc(S, omega, wvno); sl_energy(S, omega, wvno);
    for (int i = 0; i < nlayer; i++) { b2[i] = S->dcdb[i]; r2[i] = S->dcdr[i]; h2[i] = S->dcdh[i]; }
    double uc1 = *cg / *cp;
    for (int i = 0; i < nlayer; i++) {
        du2db[i] = uc1 * (2.0 - uc1) * S->dcdb[i] - uc1 * uc1 * t * (b2[i] - b1[i]) / (t2 - t1);
        du2dr[i] = uc1 * (2.0 - uc1) * S->dcdr[i] - uc1 * uc1 * t * (r2[i] - r1[i]) / (t2 - t1);
        du2dh[i] = uc1 * (2.0 - uc1) * S->dcdh[i] - uc1 * uc1 * t * (h2[i] - h1[i]) / (t2 - t1);
    }
    if (iflsph > 0) {
        double ar = 6371.0;
        omega = twopi / t;
        double tm = sqrt(1. + (3.0 * *cp / (2. * ar * omega)) * (3.0 * *cp / (2. * ar * omega)));
        double tm1 = (1.5 / (ar * omega)) * (1.5 / (ar * omega)) / tm;
        for (int i = 0; i < nlayer; i++) {
            du2db[i] = (tm * du2db[i] + *cg * *cp * dc2db[i] * tm1) * S->vtp[i];
            du2dr[i] = (tm * du2dr[i] + *cg * *cp * dc2dr[i] * tm1) * S->rtp[i];
            du2dh[i] = (tm * du2dh[i] + *cg * *cp * dc2dh[i] * tm1) * S->dtp[i];
            dc2db[i] = dc2db[i] / (tm * tm * tm) * S->vtp[i];
            dc2dr[i] = dc2dr[i] / (tm * tm * tm) * S->rtp[i];
            dc2dh[i] = dc2dh[i] / (tm * tm * tm) * S->dtp[i];
        }
        *cp = *cp / tm; *cg = *cg * tm;
    }
    sr_suffix_sum(dc2dh, nlayer);
    sr_suffix_sum(du2dh, nlayer);
    free(w); free(S);
}

/* surfdisp.cpp:151-173  _RayleighGroup: Rc roots, then sregn96's analytic U per period */
int orc_rayleigh_group(const float *thk, const float *vp, const float *vs, const float *rho,
                       int nlayer, const double *t, double *cg, int kmax)
{
    double *cp = (double *)calloc((size_t)kmax, sizeof(double));
    double *w = (double *)calloc((size_t)4 * nlayer, sizeof(double));
    int ierr = orc_surfdisp_rc(thk, vp, vs, rho, nlayer, t, cp, kmax, NULL);
    if (ierr != 1) {
        for (int i = 0; i < kmax; i++)
            orc_sregn96(thk, vp, vs, rho, nlayer, t[i], cp[i], &cg[i], NULL, NULL, NULL, NULL,
                        w, w + nlayer, w + 2 * nlayer, w + 3 * nlayer);
    }
    free(cp); free(w);
    return ierr;
}

/*
 * General entries for all four wavetypes (0 Rc, 1 Rg, 2 Lc, 3 Lg) and sphere 0/1:
 *   orc_swd_forward = libsurf.forward  (src/SWD/main.cpp:14-59: _surfdisp keep_flat=false, _RayleighGroup,
 *                     _LoveGroup surfdisp.cpp:119-173)
 *   orc_swd_kernel  = libsurf.adjoint_kernel (main.cpp:61-82 -> _SurfKernel surfdisp.cpp:190-297);
 *                     Love leaves dcda untouched in the reference (uninitialised memory): zeros here.
 */
int orc_swd_forward(const float *thk, const float *vp, const float *vs, const float *rho, int nlayer,
                    const double *t, double *cg, int kmax, int wavetype, int sphere)
{
    if (wavetype == 0) return sd_surfdisp(thk, vp, vs, rho, nlayer, t, cg, kmax, 2, sphere, 0, NULL);
    if (wavetype == 2) return sd_surfdisp(thk, vp, vs, rho, nlayer, t, cg, kmax, 1, sphere, 0, NULL);
    double *cp = (double *)calloc((size_t)kmax, sizeof(double));
    double *w = (double *)calloc((size_t)4 * nlayer, sizeof(double));
    int ierr;
    if (wavetype == 1) {
        ierr = sd_surfdisp(thk, vp, vs, rho, nlayer, t, cp, kmax, 2, sphere, 1, NULL);
        if (ierr != 1)
            for (int i = 0; i < kmax; i++)
                orc_sregn96s(thk, vp, vs, rho, nlayer, t[i], &cp[i], &cg[i], w, w + nlayer, w + 2 * nlayer, w + 3 * nlayer, sphere);
    } else {
        float *vpl = (float *)calloc((size_t)(nlayer > kmax ? nlayer : kmax), sizeof(float));
        for (int i = 0; i < nlayer; i++) vpl[i] = (float)(1.732 * vs[i]);      /* surfdisp.cpp:132 */
        ierr = sd_surfdisp(thk, vpl, vs, rho, nlayer, t, cp, kmax, 1, sphere, 1, NULL);
        if (ierr != 1)
            for (int i = 0; i < kmax; i++)
                orc_slegn96(thk, vs, rho, nlayer, t[i], &cp[i], &cg[i], w, w + nlayer, w + 2 * nlayer, sphere);
        free(vpl);
    }
    free(cp); free(w);
    return ierr;
}

int orc_swd_kernel(const float *thk, const float *vp, const float *vs, const float *rho, int nlayer,
                   const double *t, double *c, int nt, double *dcda, double *dcdb, double *dcdr,
                   double *dcdh, int wavetype, int sphere)
{
    int ierr;
    int iwave = (wavetype < 2) ? 2 : 1;
    if (wavetype == 0 || wavetype == 2) {
        ierr = sd_surfdisp(thk, vp, vs, rho, nlayer, t, c, nt, iwave, sphere, 1, NULL);
        if (ierr == 1) return ierr;
        for (int i = 0; i < nt; i++) {
            int k = i * nlayer;
            double cg;
            if (wavetype == 0) orc_sregn96s(thk, vp, vs, rho, nlayer, t[i], &c[i], &cg, dcda + k, dcdb + k, dcdh + k, dcdr + k, sphere);
            else { orc_slegn96(thk, vs, rho, nlayer, t[i], &c[i], &cg, dcdb + k, dcdh + k, dcdr + k, sphere);
                   for (int j = 0; j < nlayer; j++) dcda[k + j] = 0.0; }
        }
        return ierr;
    }
    double *buf = (double *)calloc((size_t)5 * nt + (size_t)4 * nlayer, sizeof(double));
    double *cp = buf, *cp1 = buf + nt, *cp2 = buf + 2 * nt, *t1 = buf + 3 * nt, *t2 = buf + 4 * nt, *tmp = buf + 5 * nt;
    for (int i = 0; i < nt; i++) { t1[i] = t[i] * (1.0 + 0.05); t2[i] = t[i] * (1.0 - 0.05); }
    ierr = sd_surfdisp(thk, vp, vs, rho, nlayer, t, cp, nt, iwave, sphere, 1, NULL);
    int ierr1 = sd_surfdisp(thk, vp, vs, rho, nlayer, t1, cp1, nt, iwave, sphere, 1, NULL);
    int ierr2 = sd_surfdisp(thk, vp, vs, rho, nlayer, t2, cp2, nt, iwave, sphere, 1, NULL);
    ierr = (wavetype == 1) ? ((ierr + ierr1 + ierr2) > 0) : (ierr || ierr1 || ierr2);
    if (ierr != 1) {
        for (int i = 0; i < nt; i++) {
            int k = i * nlayer;
            if (wavetype == 1)
                orc_sregnpus(thk, vp, vs, rho, nlayer, t[i], &cp[i], &c[i], t1[i], cp1[i], t2[i], cp2[i],
                             tmp, tmp + nlayer, tmp + 2 * nlayer, tmp + 3 * nlayer,
                             dcda + k, dcdb + k, dcdh + k, dcdr + k, sphere);
            else {
                orc_slegnpu(thk, vs, rho, nlayer, t[i], &cp[i], &c[i], t1[i], cp1[i], t2[i], cp2[i],
                            tmp, tmp + nlayer, tmp + 2 * nlayer, dcdb + k, dcdh + k, dcdr + k, sphere);
                for (int j = 0; j < nlayer; j++) dcda[k + j] = 0.0;
            }
        }
    }
    free(buf);
    return ierr;
}

/* the same with libsurf's `mode` argument (0 fundamental, 1 first higher mode, ...: src/SWD/main.cpp:16,63) */
int orc_swd_forward_m(const float *thk, const float *vp, const float *vs, const float *rho, int nlayer,
                      const double *t, double *cg, int kmax, int wavetype, int sphere, int mode)
{
    g_nmode = mode + 1;
    int ierr = orc_swd_forward(thk, vp, vs, rho, nlayer, t, cg, kmax, wavetype, sphere);
    g_nmode = 1;
    return ierr;
}
int orc_swd_kernel_m(const float *thk, const float *vp, const float *vs, const float *rho, int nlayer,
                     const double *t, double *c, int nt, double *dcda, double *dcdb, double *dcdr,
                     double *dcdh, int wavetype, int sphere, int mode)
{
    g_nmode = mode + 1;
    int ierr = orc_swd_kernel(thk, vp, vs, rho, nlayer, t, c, nt, dcda, dcdb, dcdr, dcdh, wavetype, sphere);
    g_nmode = 1;
    return ierr;
}

/*
 * surfdisp.cpp:190-297  _SurfKernel for wavetype 0 = "Rc", 1 = "Rg" (flat).
 * Outputs row-major [nt][nlayer] like the pybind11 wrapper (main.cpp:61-82).
 * Returns ierr (1 = root search failed).
 */
int orc_surf_kernel(const float *thk, const float *vp, const float *vs, const float *rho,
                    int nlayer, const double *t, double *c, int nt, double *dcda,
                    double *dcdb, double *dcdr, double *dcdh, int wavetype, long *nsec)
{
    int ierr;
    if (wavetype == 0) {
        ierr = orc_surfdisp_rc(thk, vp, vs, rho, nlayer, t, c, nt, nsec);
        if (ierr == 1) return ierr;
        for (int i = 0; i < nt; i++) {
            int k = i * nlayer;
            double cg;
            orc_sregn96(thk, vp, vs, rho, nlayer, t[i], c[i], &cg, NULL, NULL, NULL, NULL,
                        dcda + k, dcdb + k, dcdh + k, dcdr + k);
        }
        return ierr;
    }
    double *buf = (double *)calloc((size_t)5 * nt + (size_t)4 * nlayer, sizeof(double));
    double *cp = buf, *cp1 = buf + nt, *cp2 = buf + 2 * nt, *t1 = buf + 3 * nt, *t2 = buf + 4 * nt;
    double *tmp = buf + 5 * nt;
    for (int i = 0; i < nt; i++) {
        t1[i] = t[i] * (1.0 + 0.05);
        t2[i] = t[i] * (1.0 - 0.05);
    }
    ierr = orc_surfdisp_rc(thk, vp, vs, rho, nlayer, t, cp, nt, nsec);
    int ierr1 = orc_surfdisp_rc(thk, vp, vs, rho, nlayer, t1, cp1, nt, nsec);
    int ierr2 = orc_surfdisp_rc(thk, vp, vs, rho, nlayer, t2, cp2, nt, nsec);
    ierr = (ierr + ierr1 + ierr2) > 0;
    if (ierr != 1) {
        for (int i = 0; i < nt; i++) {
            int k = i * nlayer;
            orc_sregnpu(thk, vp, vs, rho, nlayer, t[i], cp[i], &c[i], t1[i], cp1[i], t2[i], cp2[i],
                        tmp, tmp + nlayer, tmp + 2 * nlayer, tmp + 3 * nlayer,
                        dcda + k, dcdb + k, dcdh + k, dcdr + k);
        }
    }
    free(buf);
    return ierr;
}
