/*
 * oracle/rf_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's frequency-domain receiver-function
 * path, one function per reference routine (file:line under /root/reference):
 *
 *   src/RF/RFModule.f90   cal_rf_freq :193-255, cal_rf_par_freq_all :343-430,
 *                         cal_response :432-478, cal_response_par_all :592-707,
 *                         cal_matrix_a :709-764, cal_matrix_a_par :766-879,
 *                         cal_E_inv :881-922, cal_E_inv_par :924-987
 *   src/RF/deconit.f90    nextpow2 :1-13
 *   src/RF/fftpack.f90    irfft :23-42  (FFTW3 c2r + 1/n)
 *
 * The propagator/partials core follows the reference's own O(nlayer^2) loop so
 * that it can be checked routine by routine against the compiled reference
 * (oracle/_ref/librf_core_ref.so through oracle/ref_probe.c).  The FFT is the
 * one piece with no reference build behind it (FFTW3, pinned ">=3.3" in the
 * reference README, is not in this image): irfft below restates FFTW's
 * documented c2r semantics -- unnormalised inverse of a Hermitian half-spectrum
 * in which the imaginary parts of the DC and Nyquist bins are ignored -- and is
 * cross-checked against numpy.fft.irfft in tests/.  End-to-end RF values are
 * additionally anchored to the known answers recorded in SURVEY.md section 8(c).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this file's shared object; the product never does.
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef double _Complex zc;

/* float32 pi: `pi = atan(1.0) * 4.0` in default real (RFModule.f90:211,278,364) */
static const double RF_PI = (double)3.14159274101257324f;

/* deconit.f90:1-13 */
int orc_nextpow2(int n)
{
    int nout = 1;
    while (nout < n) nout = nout * 2;
    return nout;
}

/* c = a*b for column-major 4x4 complex (Fortran matmul), explicit real arithmetic */
static void mat4_mul(const zc *a, const zc *b, zc *c)
{
    zc t[16];
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++) {
            double sr = 0.0, si = 0.0;
            for (int k = 0; k < 4; k++) {
                double ar = creal(a[k * 4 + i]), ai = cimag(a[k * 4 + i]);
                double br = creal(b[j * 4 + k]), bi = cimag(b[j * 4 + k]);
                sr += ar * br - ai * bi;
                si += ar * bi + ai * br;
            }
            t[j * 4 + i] = sr + si * I;
        }
    memcpy(c, t, sizeof(t));
}
#define M(a, i, j) (a)[((j) - 1) * 4 + ((i) - 1)] /* 1-based Fortran a(i,j) */

typedef struct {
    zc k, miu, v_alpha, v_beta, gamma, gamma1, gamma2, gamma3, va_k, vb_k;
    zc c_a, x_a, y_a, c_b, x_b, y_b;
} lay_terms;

/* common sub-expressions of cal_matrix_a / cal_matrix_a_par (:721-744, :781-802) */
static void layer_terms(zc omega, double ray_p, double thick, zc alpha, zc beta,
                        double rho, lay_terms *L)
{
    L->miu = rho * (beta * beta);
    L->k = omega * ray_p;
    zc k_alpha = omega / alpha, k_beta = omega / beta;
    L->v_alpha = csqrt(L->k * L->k - k_alpha * k_alpha);
    L->v_beta = csqrt(L->k * L->k - k_beta * k_beta);
    L->va_k = csqrt(ray_p * ray_p - 1.0 / (alpha * alpha)) / ray_p;
    L->vb_k = csqrt(ray_p * ray_p - 1.0 / (beta * beta)) / ray_p;
    L->gamma = 2 * (ray_p * ray_p) * (beta * beta);
    L->gamma1 = 1 - 1 / L->gamma;
    L->gamma2 = L->gamma / ((alpha * ray_p) * (alpha * ray_p));
    L->gamma3 = 1. / (L->gamma - 2);
    L->c_a = ccosh(L->v_alpha * thick);
    L->x_a = L->va_k * csinh(L->v_alpha * thick);
    L->y_a = csinh(L->v_alpha * thick) / L->va_k;
    L->c_b = ccosh(L->v_beta * thick);
    L->x_b = L->vb_k * csinh(L->v_beta * thick);
    L->y_b = csinh(L->v_beta * thick) / L->vb_k;
}

/* RFModule.f90:709-764 */
void orc_rf_matrix_a(zc omega, double ray_p, double thick, zc alpha, zc beta,
                     double rho, zc *a)
{
    lay_terms L;
    layer_terms(omega, ray_p, thick, alpha, beta, rho, &L);
    zc g1 = L.gamma1, miu = L.miu;
    zc c_a = L.c_a, x_a = L.x_a, y_a = L.y_a, c_b = L.c_b, x_b = L.x_b, y_b = L.y_b;
    M(a, 1, 1) = c_a - g1 * c_b;
    M(a, 1, 2) = g1 * y_a - x_b;
    M(a, 1, 3) = (c_b - c_a) / 2 / miu;
    M(a, 1, 4) = (x_b - y_a) / 2 / miu;
    M(a, 2, 1) = g1 * y_b - x_a;
    M(a, 2, 2) = c_b - g1 * c_a;
    M(a, 2, 3) = (x_a - y_b) / 2 / miu;
    M(a, 2, 4) = (c_a - c_b) / 2 / miu;
    M(a, 3, 1) = 2 * miu * g1 * (c_a - c_b);
    M(a, 3, 2) = 2 * miu * (g1 * g1 * y_a - x_b);
    M(a, 3, 3) = c_b - g1 * c_a;
    M(a, 3, 4) = x_b - g1 * y_a;
    M(a, 4, 1) = 2 * miu * (g1 * g1 * y_b - x_a);
    M(a, 4, 2) = 2 * miu * g1 * (c_b - c_a);
    M(a, 4, 3) = x_a - g1 * y_b;
    M(a, 4, 4) = c_a - g1 * c_b;
    for (int i = 0; i < 16; i++) a[i] = L.gamma * a[i];
}

/* RFModule.f90:766-879; ipars 1 rho, 2 vp, 3 vs, 4 thickness */
void orc_rf_matrix_a_par(zc omega, double ray_p, double thick, zc alpha, zc beta,
                         double rho, zc *a, int ipars)
{
    lay_terms L;
    layer_terms(omega, ray_p, thick, alpha, beta, rho, &L);
    zc k = L.k, g = L.gamma, g1 = L.gamma1, g2 = L.gamma2, g3 = L.gamma3, miu = L.miu;
    zc c_a = L.c_a, x_a = L.x_a, y_a = L.y_a, c_b = L.c_b, x_b = L.x_b, y_b = L.y_b;
    zc va_k = L.va_k, vb_k = L.vb_k, v_alpha = L.v_alpha, v_beta = L.v_beta;
    if (ipars == 3) {
        M(a, 1, 1) = 2. / beta * (g * (c_a - c_b) - g1 * k * thick * y_b);
        M(a, 1, 2) = 2. / beta * (g * (y_a - x_b) - (k * thick * c_b + y_b));
        M(a, 1, 3) = k * thick * y_b / miu / beta;
        M(a, 1, 4) = (k * thick * c_b + y_b) / miu / beta;
        M(a, 2, 1) = ((y_b - x_a) + g1 * g3 * (k * thick * c_b - y_b)) * 2 * g / beta;
        M(a, 2, 2) = 2. / beta * (g * (c_b - c_a) + k * thick * y_b);
        M(a, 2, 3) = -(k * thick * c_b - y_b) * g * g3 / miu / beta;
        M(a, 2, 4) = -M(a, 1, 3);
        M(a, 3, 1) = 4. * miu / beta * ((2 * g - 1) * (c_a - c_b) - g1 * k * thick * y_b);
        M(a, 3, 2) = 4. * miu / beta * ((2 * g) * (g1 * y_a - x_b) - (k * thick * c_b + y_b));
        M(a, 3, 3) = M(a, 2, 2);
        M(a, 3, 4) = -M(a, 1, 2);
        M(a, 4, 1) = 4. * miu * g / beta * (2 * g1 * y_b - 2 * x_a + g1 * g1 * g3 * (k * thick * c_b - y_b));
        M(a, 4, 2) = -M(a, 3, 1);
        M(a, 4, 3) = -M(a, 2, 1);
        M(a, 4, 4) = M(a, 1, 1);
    } else if (ipars == 2) {
        zc va2 = va_k * va_k;
        M(a, 1, 1) = k * thick * y_a * g2 / alpha;
        M(a, 1, 2) = 1 / va2 / alpha * g1 * g2 * (k * thick * c_a - y_a);
        M(a, 1, 3) = -k * thick * y_a * g2 / 2 / miu / alpha;
        M(a, 1, 4) = -1 / va2 * (k * thick * c_a - y_a) * g2 / 2 / miu / alpha;
        M(a, 2, 1) = -(k * thick * c_a + y_a) * g2 / alpha;
        M(a, 2, 2) = -k * thick * y_a * g1 * g2 / alpha;
        M(a, 2, 3) = (k * thick * c_a + y_a) * g2 / 2 / miu / alpha;
        M(a, 2, 4) = k * thick * y_a * g2 / 2 / miu / alpha;
        M(a, 3, 1) = k * thick * y_a * g1 * g2 * 2 * miu / alpha;
        M(a, 3, 2) = 2. * miu / alpha * (g1 * g1) * g2 * (k * thick * c_a - y_a) / va2;
        M(a, 3, 3) = -k * thick * y_a * g1 * g2 / alpha;
        M(a, 3, 4) = -1. / alpha / va2 * (k * thick * c_a - y_a) * g1 * g2;
        M(a, 4, 1) = -2. * miu / alpha * (k * thick * c_a + y_a) * g2;
        M(a, 4, 2) = -2. * miu / alpha * k * thick * y_a * g1 * g2;
        M(a, 4, 3) = (k * thick * c_a + y_a) * g2 / alpha;
        M(a, 4, 4) = k * thick * y_a / alpha * g2;
    } else if (ipars == 1) {
        for (int i = 0; i < 16; i++) a[i] = 0.0;
        M(a, 1, 3) = -g / (2 * rho * miu) * (-c_a + c_b);
        M(a, 1, 4) = -g / (2 * rho * miu) * (-y_a + x_b);
        M(a, 2, 3) = -g / (2 * rho * miu) * (x_a - y_b);
        M(a, 2, 4) = -g / (2 * rho * miu) * (c_a - c_b);
        M(a, 3, 1) = 2. * miu * g * g1 / rho * (c_a - c_b);
        M(a, 3, 2) = 2. * miu * g / rho * (g1 * g1 * y_a - x_b);
        M(a, 4, 1) = 2. * miu * g / rho * (-x_a + g1 * g1 * y_b);
        M(a, 4, 2) = 2. * miu * g * g1 / rho * (-c_a + c_b);
    } else {
        M(a, 1, 1) = (x_a - g1 * x_b) * k;
        M(a, 1, 2) = g1 * k * c_a - v_beta * vb_k * c_b;
        M(a, 1, 3) = (x_b - x_a) * k / 2 / miu;
        M(a, 1, 4) = (v_beta * vb_k * c_b - k * c_a) / 2 / miu;
        M(a, 2, 1) = g1 * k * c_b - v_alpha * va_k * c_a;
        M(a, 2, 2) = (x_b - g1 * x_a) * k;
        M(a, 2, 3) = (v_alpha * va_k * c_a - k * c_b) / 2 / miu;
        M(a, 2, 4) = (x_a - x_b) * k / 2 / miu;
        M(a, 3, 1) = 2. * miu * g1 * k * (x_a - x_b);
        M(a, 3, 2) = 2. * miu * (g1 * g1 * k * c_a - v_beta * vb_k * c_b);
        M(a, 3, 3) = (x_b - g1 * x_a) * k;
        M(a, 3, 4) = v_beta * vb_k * c_b - g1 * k * c_a;
        M(a, 4, 1) = 2. * miu * (g1 * g1 * k * c_b - v_alpha * va_k * c_a);
        M(a, 4, 2) = 2. * miu * g1 * k * (x_b - x_a);
        M(a, 4, 3) = v_alpha * va_k * c_a - g1 * k * c_b;
        M(a, 4, 4) = (x_a - g1 * x_b) * k;
        for (int i = 0; i < 16; i++) a[i] = g * a[i];
    }
}

/* RFModule.f90:881-922 */
void orc_rf_e_inv(zc omega, double ray_p, zc alpha, zc beta, double rho, zc *e)
{
    (void)omega;
    zc miu = rho * (beta * beta);
    zc gamma = 2 * (ray_p * ray_p) * (beta * beta);
    zc gamma1 = 1 - 1 / gamma;
    zc va_k = csqrt(ray_p * ray_p - 1.0 / (alpha * alpha)) / ray_p;
    zc vb_k = csqrt(ray_p * ray_p - 1.0 / (beta * beta)) / ray_p;
    M(e, 1, 1) = -1.0;
    M(e, 1, 2) = -gamma1 / va_k;
    M(e, 1, 3) = 1.0 / (2. * miu);
    M(e, 1, 4) = 1 / (2. * miu * va_k);
    M(e, 2, 1) = gamma1 / vb_k;
    M(e, 2, 2) = 1.0;
    M(e, 2, 3) = -1 / (2. * miu * vb_k);
    M(e, 2, 4) = -1.0 / (2. * miu);
    M(e, 3, 1) = 1.0;
    M(e, 3, 2) = -gamma1 / va_k;
    M(e, 3, 3) = -1.0 / (2. * miu);
    M(e, 3, 4) = 1 / (2. * miu * va_k);
    M(e, 4, 1) = -gamma1 / vb_k;
    M(e, 4, 2) = 1.0;
    M(e, 4, 3) = 1. / (2. * miu * vb_k);
    M(e, 4, 4) = -1.0 / (2. * miu);
    for (int i = 0; i < 16; i++) e[i] = e[i] * 0.5 * gamma;
}

/*
 * RFModule.f90:924-987.  For ipars == 2 the reference multiplies rows 1 and 3 by
 * an UNASSIGNED local va_k (:933,980): those rows are undefined in the reference.
 * Row 2 (the one a P-type RF reads, :653-655) is identically zero either way.
 * Here rows 1,3 use the value the author evidently meant (sqrt(p^2-1/alpha^2)/p);
 * nothing on the P path depends on it.
 */
void orc_rf_e_inv_par(zc omega, double ray_p, zc alpha, zc beta, double rho, zc *e, int ipars)
{
    zc miu = rho * (beta * beta);
    zc k = omega * ray_p;
    zc k_alpha = omega / alpha, k_beta = omega / beta;
    zc v_alpha = csqrt(k * k - k_alpha * k_alpha);
    zc v_beta = csqrt(k * k - k_beta * k_beta);
    zc gamma = 2 * (k * k) * (beta * beta) / (omega * omega);
    zc gamma1 = 1 - 1 / gamma;
    zc gamma3 = 1.0 / (gamma - 2);
    for (int i = 0; i < 16; i++) e[i] = 0.0;
    if (ipars == 3) {
        M(e, 1, 1) = -1.0;
        M(e, 1, 2) = -k / v_alpha;
        M(e, 2, 1) = k / v_beta * (1 - gamma1 * gamma3);
        M(e, 2, 2) = 1.0;
        M(e, 2, 3) = k * gamma3 / 2 / miu / v_beta;
        M(e, 3, 1) = 1.0;
        M(e, 3, 2) = -k / v_alpha;
        M(e, 4, 1) = -k / v_beta * (1 - gamma3 * gamma1);
        M(e, 4, 2) = 1.0;
        M(e, 4, 3) = -k * gamma3 / 2 / miu / v_beta;
        for (int i = 0; i < 16; i++) e[i] = e[i] * gamma / beta;
    } else if (ipars == 1) {
        M(e, 1, 3) = -1.0;
        M(e, 1, 4) = -k / v_alpha;
        M(e, 2, 3) = k / v_beta;
        M(e, 2, 4) = 1.0;
        M(e, 3, 3) = 1.0;
        M(e, 3, 4) = -k / v_alpha;
        M(e, 4, 3) = -k / v_beta;
        M(e, 4, 4) = 1.0;
        for (int i = 0; i < 16; i++) e[i] = e[i] * gamma / 4.0 / rho / miu;
    } else if (ipars == 2) {
        zc va_k = csqrt(ray_p * ray_p - 1.0 / (alpha * alpha)) / ray_p;
        M(e, 1, 2) = gamma1;
        M(e, 1, 4) = -0.5 / miu;
        M(e, 3, 2) = gamma1;
        M(e, 3, 4) = -0.5 / miu;
        for (int i = 0; i < 16; i++)
            e[i] = e[i] * (beta * beta) / (alpha * alpha * alpha) / (va_k * va_k * va_k);
    }
}

static void mat4_eye(zc *a)
{
    for (int i = 0; i < 16; i++) a[i] = 0.0;
    for (int i = 0; i < 4; i++) a[i * 4 + i] = 1.0;
}

/* RFModule.f90:432-478 */
int orc_rf_response(zc omega, double ray_p, const double *thk, const zc *alpha,
                    const zc *beta, const double *rho, int nlayer, int rf_type,
                    zc *R21, zc *R22)
{
    zc a_syn[16], a1[16], einv[16];
    mat4_eye(a_syn);
    for (int ilayer = 1; ilayer <= nlayer - 1; ilayer++) {
        int inv = nlayer - ilayer - 1; /* 0-based ilayer_inv */
        orc_rf_matrix_a(omega, ray_p, thk[inv], alpha[inv], beta[inv], rho[inv], a1);
        mat4_mul(a_syn, a1, a_syn);
    }
    orc_rf_e_inv(omega, ray_p, alpha[nlayer - 1], beta[nlayer - 1], rho[nlayer - 1], einv);
    mat4_mul(einv, a_syn, a_syn);
    if (rf_type == 1) { *R22 = M(a_syn, 2, 2) * I; *R21 = M(a_syn, 2, 1); }
    else if (rf_type == 2) { *R22 = -M(a_syn, 1, 1) * I; *R21 = M(a_syn, 1, 2); }
    else return -1;
    return 0;
}

static zc nan_scrub(zc z) { return isnan(cabs(z)) ? 0.0 : z; }

/* RFModule.f90:592-707; R21_m/R22_m are Fortran (nlayer, 4): [ipar*nlayer + layer] */
void orc_rf_response_par_all(zc omega, double ray_p, const double *thk, const zc *alpha,
                             const zc *beta, const double *vp, const double *vs,
                             const double *rho, int nlayer, int rf_type, zc *R21, zc *R22,
                             zc *R21_m, zc *R22_m)
{
    zc *all_a = (zc *)malloc(sizeof(zc) * 16 * (size_t)nlayer);
    zc *all_a_m = (zc *)malloc(sizeof(zc) * 16 * 4 * (size_t)nlayer);
    zc einv[16], einv_par[4][16], a_syn[16], a_syn_m[16];
    for (int il = 0; il < nlayer - 1; il++) {
        orc_rf_matrix_a(omega, ray_p, thk[il], alpha[il], beta[il], rho[il], all_a + 16 * il);
        for (int ipar = 1; ipar <= 4; ipar++) {
            zc *am = all_a_m + 16 * (4 * il + ipar - 1);
            orc_rf_matrix_a_par(omega, ray_p, thk[il], alpha[il], beta[il], rho[il], am, ipar);
            if (ipar == 3) for (int i = 0; i < 16; i++) am[i] = am[i] * beta[il] / vs[il];
            else if (ipar == 2) for (int i = 0; i < 16; i++) am[i] = am[i] * alpha[il] / vp[il];
        }
    }
    int nl = nlayer - 1;
    orc_rf_e_inv(omega, ray_p, alpha[nl], beta[nl], rho[nl], einv);
    for (int ipar = 1; ipar <= 4; ipar++) {
        zc *ep = einv_par[ipar - 1];
        orc_rf_e_inv_par(omega, ray_p, alpha[nl], beta[nl], rho[nl], ep, ipar);
        if (ipar == 3) for (int i = 0; i < 16; i++) ep[i] = ep[i] * beta[nl] / vs[nl];
        else if (ipar == 2) for (int i = 0; i < 16; i++) ep[i] = ep[i] * alpha[nl] / vp[nl];
    }
    /* the reference repeats this product nlayer times (:644-650); once is the same */
    mat4_eye(a_syn);
    for (int ilayer = 1; ilayer <= nlayer - 1; ilayer++)
        mat4_mul(a_syn, all_a + 16 * (nlayer - ilayer - 1), a_syn);
    mat4_mul(einv, a_syn, a_syn);
    if (rf_type == 1) { *R22 = M(a_syn, 2, 2) * I; *R21 = M(a_syn, 2, 1); }
    else { *R22 = -M(a_syn, 1, 1) * I; *R21 = M(a_syn, 1, 2); }
    *R22 = nan_scrub(*R22);
    *R21 = nan_scrub(*R21);
    for (int ipar = 1; ipar <= 4; ipar++) {
        for (int par_layer = 0; par_layer < nlayer; par_layer++) {
            mat4_eye(a_syn_m);
            for (int ilayer = 1; ilayer <= nlayer - 1; ilayer++) {
                int inv = nlayer - ilayer - 1;
                if (par_layer == inv) mat4_mul(a_syn_m, all_a_m + 16 * (4 * inv + ipar - 1), a_syn_m);
                else mat4_mul(a_syn_m, all_a + 16 * inv, a_syn_m);
            }
            if (par_layer == nlayer - 1) mat4_mul(einv_par[ipar - 1], a_syn_m, a_syn_m);
            else mat4_mul(einv, a_syn_m, a_syn_m);
            zc r22, r21;
            if (rf_type == 1) { r22 = M(a_syn_m, 2, 2) * I; r21 = M(a_syn_m, 2, 1); }
            else { r22 = -M(a_syn_m, 1, 1) * I; r21 = M(a_syn_m, 1, 2); }
            R22_m[(ipar - 1) * nlayer + par_layer] = nan_scrub(r22);
            R21_m[(ipar - 1) * nlayer + par_layer] = nan_scrub(r21);
        }
    }
    free(all_a); free(all_a_m);
}

/* in-place radix-2 complex FFT, sign = +1 -> exp(+i...), n a power of two */
static void fft_pow2(zc *x, int n, int sign)
{
    for (int i = 1, j = 0; i < n; i++) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { zc t = x[i]; x[i] = x[j]; x[j] = t; }
    }
    for (int len = 2; len <= n; len <<= 1) {
        double ang = sign * 2.0 * 3.14159265358979323846 / len;
        for (int i = 0; i < n; i += len)
            for (int k = 0; k < len / 2; k++) {
                zc w = cos(ang * k) + sin(ang * k) * I;
                zc u = x[i + k], v = x[i + k + len / 2] * w;
                x[i + k] = u + v; x[i + k + len / 2] = u - v;
            }
    }
}

/* fftpack.f90:23-42: FFTW c2r (imaginary parts of DC and Nyquist ignored), then /n */
void orc_irfft(const zc *inp, double *out, int n)
{
    zc *x = (zc *)malloc(sizeof(zc) * (size_t)n);
    x[0] = creal(inp[0]);
    x[n / 2] = creal(inp[n / 2]);
    for (int k = 1; k < n / 2; k++) { x[k] = inp[k]; x[n - k] = conj(inp[k]); }
    fft_pow2(x, n, +1);
    for (int i = 0; i < n; i++) out[i] = creal(x[i]) / n;
    free(x);
}

static void attenuate(const double *v, const double *q, int n, zc *out)
{
    for (int i = 0; i < n; i++)
        out[i] = v[i] * (1.0 + I / (2.0 * q[i]) + 1.0 / (8.0 * (q[i] * q[i])));
}

static void spectrum_to_trace(const zc *spec, double *tmp, double *out, int nft, int nt,
                              double dt, double sigma, double t0)
{
    orc_irfft(spec, tmp, nft);
    for (int it = 1; it <= nt; it++)
        out[it - 1] = tmp[it - 1] / dt * exp(sigma * (-t0 + (it - 1) * dt));
}

/* RFModule.f90:193-255  cal_rf_freq */
int orc_rf_freq(const double *thk, const double *vp, const double *vs, const double *rho,
                const double *qa, const double *qb, int nlayer, int nt, double dt,
                double ray_p, double f0, double t0, double water, int rf_type, double *rcv_fun)
{
    int nft = orc_nextpow2(nt), n2 = nft / 2 + 1;
    zc *alpha = (zc *)malloc(sizeof(zc) * (size_t)nlayer), *beta = (zc *)malloc(sizeof(zc) * (size_t)nlayer);
    zc *R21 = (zc *)malloc(sizeof(zc) * (size_t)n2), *R22 = (zc *)malloc(sizeof(zc) * (size_t)n2);
    zc *spec = (zc *)malloc(sizeof(zc) * (size_t)n2);
    double *w = (double *)malloc(sizeof(double) * (size_t)n2), *wa = (double *)malloc(sizeof(double) * (size_t)n2);
    double *tmp = (double *)malloc(sizeof(double) * (size_t)nft);
    attenuate(vp, qa, nlayer, alpha);
    attenuate(vs, qb, nlayer, beta);
    double sigma = 1.0 / (dt) / nft * 4;
    int rc = 0;
    for (int it = 1; it <= n2; it++) {
        w[it - 1] = (1.0 / dt / nft) * (it - 1) * 2 * RF_PI;
        zc omega = w[it - 1] - sigma * I;
        rc |= orc_rf_response(omega, ray_p, thk, alpha, beta, rho, nlayer, rf_type, &R21[it - 1], &R22[it - 1]);
    }
    double wmax = -INFINITY;
    for (int i = 0; i < n2; i++) { wa[i] = creal(R21[i] * conj(R21[i])); if (wa[i] > wmax) wmax = wa[i]; }
    for (int i = 0; i < n2; i++) {
        double gauss = exp(-((w[i] / 2 / f0) * (w[i] / 2 / f0)));
        double fai = fmax(wa[i], water * wmax);
        spec[i] = conj(R21[i]) * R22[i] * gauss * cexp(-I * w[i] * t0) / fai;
    }
    spectrum_to_trace(spec, tmp, rcv_fun, nft, nt, dt, sigma, t0);
    free(alpha); free(beta); free(R21); free(R22); free(spec); free(w); free(wa); free(tmp);
    return rc;
}

/*
 * RFModule.f90:343-430  cal_rf_par_freq_all.  rcv_fun_p is the C view of the
 * Fortran array (nt, nlayer, 4): [ipar][layer][nt], ipar 0..3 = rho, vp, vs, thk
 * (src/RF/main.cpp:140-189 returns exactly this buffer as k[4,nlayer,nt]).
 * If spec_out != NULL it receives R21, R22 (n2 each) then R21_m, R22_m as
 * [it][ipar][layer] for routine-level comparison with the compiled reference.
 */
int orc_rf_par_freq_all(const double *thk, const double *vp, const double *vs,
                        const double *rho, const double *qa, const double *qb, int nlayer,
                        int nt, double dt, double ray_p, double f0, double t0, double water,
                        int rf_type, double *rcv_fun, double *rcv_fun_p)
{
    int nft = orc_nextpow2(nt), n2 = nft / 2 + 1, np = 4 * nlayer;
    zc *alpha = (zc *)malloc(sizeof(zc) * (size_t)nlayer), *beta = (zc *)malloc(sizeof(zc) * (size_t)nlayer);
    zc *R21 = (zc *)malloc(sizeof(zc) * (size_t)n2), *R22 = (zc *)malloc(sizeof(zc) * (size_t)n2);
    zc *R21_m = (zc *)malloc(sizeof(zc) * (size_t)n2 * np), *R22_m = (zc *)malloc(sizeof(zc) * (size_t)n2 * np);
    zc *spec = (zc *)malloc(sizeof(zc) * (size_t)n2), *R21sq = (zc *)malloc(sizeof(zc) * (size_t)n2);
    double *w = (double *)malloc(sizeof(double) * (size_t)n2), *wa = (double *)malloc(sizeof(double) * (size_t)n2);
    double *fai = (double *)malloc(sizeof(double) * (size_t)n2), *gauss = (double *)malloc(sizeof(double) * (size_t)n2);
    double *tmp = (double *)malloc(sizeof(double) * (size_t)nft);
    attenuate(vp, qa, nlayer, alpha);
    attenuate(vs, qb, nlayer, beta);
    double sigma = 1.0 / (dt) / nft * 4.;
    for (int it = 1; it <= n2; it++) {
        w[it - 1] = 1.0 / nft / dt * (it - 1) * 2.0 * RF_PI;
        zc omega = w[it - 1] - sigma * I;
        orc_rf_response_par_all(omega, ray_p, thk, alpha, beta, vp, vs, rho, nlayer, rf_type,
                                &R21[it - 1], &R22[it - 1], R21_m + (size_t)(it - 1) * np,
                                R22_m + (size_t)(it - 1) * np);
    }
    double wmax = -INFINITY;
    for (int i = 0; i < n2; i++) {
        gauss[i] = exp(-((w[i] / 2 / f0) * (w[i] / 2 / f0)));
        wa[i] = creal(R21[i] * conj(R21[i]));
        if (wa[i] > wmax) wmax = wa[i];
    }
    for (int i = 0; i < n2; i++) {
        fai[i] = fmax(wa[i], water * wmax);
        spec[i] = conj(R21[i]) * R22[i] * gauss[i] * cexp(-I * w[i] * t0) / fai[i];
    }
    spectrum_to_trace(spec, tmp, rcv_fun, nft, nt, dt, sigma, t0);
    wmax = -INFINITY;
    for (int i = 0; i < n2; i++) {
        R21sq[i] = R21[i] * R21[i];
        wa[i] = creal(R21sq[i] * conj(R21sq[i]));
        if (wa[i] > wmax) wmax = wa[i];
    }
    for (int i = 0; i < n2; i++) fai[i] = fmax(wa[i], water * wmax);
    for (int ipar = 0; ipar < 4; ipar++)
        for (int pl = 0; pl < nlayer; pl++) {
            for (int i = 0; i < n2; i++) {
                zc r22m = R22_m[(size_t)i * np + ipar * nlayer + pl];
                zc r21m = R21_m[(size_t)i * np + ipar * nlayer + pl];
                spec[i] = conj(R21sq[i]) * (r22m * R21[i] - r21m * R22[i]) * gauss[i] *
                          cexp(-I * w[i] * t0) / fai[i];
            }
            spectrum_to_trace(spec, tmp, rcv_fun_p + ((size_t)ipar * nlayer + pl) * nt, nft, nt, dt, sigma, t0);
        }
    free(alpha); free(beta); free(R21); free(R22); free(R21_m); free(R22_m); free(spec);
    free(R21sq); free(w); free(wa); free(fai); free(gauss); free(tmp);
    return 0;
}

/* flat-argument wrappers so that ctypes can call the per-frequency routines the
 * same way it calls oracle/ref_probe.c (complex numbers as interleaved doubles) */
void orcprobe_rf_response_par_all(double w_re, double w_im, double ray_p, int nlayer,
    const double *thk, const double *alpha, const double *beta, const double *vp,
    const double *vs, const double *rho, int rf_type,
    double *R21, double *R22, double *R21_m, double *R22_m)
{
    orc_rf_response_par_all(w_re + w_im * I, ray_p, thk, (const zc *)alpha, (const zc *)beta,
                            vp, vs, rho, nlayer, rf_type, (zc *)R21, (zc *)R22, (zc *)R21_m, (zc *)R22_m);
}

void orcprobe_rf_response(double w_re, double w_im, double ray_p, int nlayer,
    const double *thk, const double *alpha, const double *beta, const double *rho,
    int rf_type, double *R21, double *R22)
{
    orc_rf_response(w_re + w_im * I, ray_p, thk, (const zc *)alpha, (const zc *)beta, rho,
                    nlayer, rf_type, (zc *)R21, (zc *)R22);
}

void orcprobe_rf_matrix_a(double w_re, double w_im, double ray_p, double thick,
    const double *alpha, const double *beta, double rho, int ipars, double *a)
{
    zc al = alpha[0] + alpha[1] * I, be = beta[0] + beta[1] * I;
    if (ipars == 0) orc_rf_matrix_a(w_re + w_im * I, ray_p, thick, al, be, rho, (zc *)a);
    else orc_rf_matrix_a_par(w_re + w_im * I, ray_p, thick, al, be, rho, (zc *)a, ipars);
}

void orcprobe_rf_e_inv(double w_re, double w_im, double ray_p, const double *alpha,
    const double *beta, double rho, int ipars, double *e)
{
    zc al = alpha[0] + alpha[1] * I, be = beta[0] + beta[1] * I;
    if (ipars == 0) orc_rf_e_inv(w_re + w_im * I, ray_p, al, be, rho, (zc *)e);
    else orc_rf_e_inv_par(w_re + w_im * I, ray_p, al, be, rho, (zc *)e, ipars);
}
