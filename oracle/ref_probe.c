/*
 * oracle/ref_probe.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * C-callable probes into the *compiled reference* receiver-function module
 * (/root/reference/src/RF/RFModule.f90).  The reference's public entry points
 * (cal_rf_freq_, cal_rf_par_freq_all_, ...) end in irfft/rfft, thin wrappers
 * (src/RF/fftpack.f90:1-42) around FFTW3, which is absent from this image.  No
 * stand-in for FFTW is written: fftpack.f90 is simply not built, the public
 * entry points are garbage-collected at link time (ref_probe.map), and the
 * probes below reach the module procedures that hold all of the physics --
 * cal_response (RFModule.f90:432-478), cal_response_par_all (:592-707),
 * cal_matrix_a (:709-764), cal_matrix_a_par (:766-879), cal_E_inv (:881-922),
 * cal_E_inv_par (:924-987) -- through their flang module-procedure symbols
 * (_QMrfmoduleP<name>, Fortran ABI: every argument by reference, explicit-shape
 * arrays as bare pointers, column-major).
 * Round 6: refprobe_gauss_filter reaches deconit.f90:15-32 (`gauss_filter`), the one routine of the
 * time-domain path besides nextpow2 that calls neither rfft nor irfft.
 */
#include <complex.h>

typedef double _Complex zc;

extern void _QMrfmodulePcal_response(const zc *omega, const double *ray_p,
    const double *thk, const zc *alpha, const zc *beta, const double *rho,
    const int *nlayer, const int *rf_type, zc *R21, zc *R22);
extern void _QMrfmodulePcal_response_par_all(const zc *omega, const double *ray_p,
    const double *thk, const zc *alpha, const zc *beta, const double *vp,
    const double *vs, const double *rho, const int *nlayer, const int *rf_type,
    zc *R21, zc *R22, zc *R21_m, zc *R22_m);
extern void _QMrfmodulePcal_matrix_a(const zc *omega, const double *ray_p,
    const double *thick, const zc *alpha, const zc *beta, const double *rho, zc *a);
extern void _QMrfmodulePcal_matrix_a_par(const zc *omega, const double *ray_p,
    const double *thick, const zc *alpha, const zc *beta, const double *rho, zc *a,
    const int *ipars);
extern void _QMrfmodulePcal_e_inv(const zc *omega, const double *ray_p,
    const zc *alpha, const zc *beta, const double *rho, zc *e);
extern void _QMrfmodulePcal_e_inv_par(const zc *omega, const double *ray_p,
    const zc *alpha, const zc *beta, const double *rho, zc *e, const int *ipars);
extern void nextpow2_(const int *n, int *nout);
extern void gauss_filter_(const int *nt, const double *dt, const double *f0, double *gauss);

/* omega = (w_re, w_im); alpha/beta complex[nlayer] interleaved (re,im).
 * R21_m/R22_m come back Fortran-shaped (nlayer, 4): index [ipar*nlayer + layer]. */
void refprobe_rf_response(double w_re, double w_im, double ray_p, int nlayer,
    const double *thk, const double *alpha, const double *beta, const double *rho,
    int rf_type, double *R21, double *R22)
{
    zc om = w_re + w_im * I;
    _QMrfmodulePcal_response(&om, &ray_p, thk, (const zc *)alpha, (const zc *)beta,
                             rho, &nlayer, &rf_type, (zc *)R21, (zc *)R22);
}

void refprobe_rf_response_par_all(double w_re, double w_im, double ray_p, int nlayer,
    const double *thk, const double *alpha, const double *beta, const double *vp,
    const double *vs, const double *rho, int rf_type,
    double *R21, double *R22, double *R21_m, double *R22_m)
{
    zc om = w_re + w_im * I;
    _QMrfmodulePcal_response_par_all(&om, &ray_p, thk, (const zc *)alpha,
        (const zc *)beta, vp, vs, rho, &nlayer, &rf_type,
        (zc *)R21, (zc *)R22, (zc *)R21_m, (zc *)R22_m);
}

/* a: complex[16], Fortran column-major a(i,j) -> a[(j-1)*4 + (i-1)] */
void refprobe_rf_matrix_a(double w_re, double w_im, double ray_p, double thick,
    const double *alpha, const double *beta, double rho, int ipars, double *a)
{
    zc om = w_re + w_im * I;
    zc al = alpha[0] + alpha[1] * I, be = beta[0] + beta[1] * I;
    if (ipars == 0)
        _QMrfmodulePcal_matrix_a(&om, &ray_p, &thick, &al, &be, &rho, (zc *)a);
    else
        _QMrfmodulePcal_matrix_a_par(&om, &ray_p, &thick, &al, &be, &rho, (zc *)a, &ipars);
}

void refprobe_rf_e_inv(double w_re, double w_im, double ray_p,
    const double *alpha, const double *beta, double rho, int ipars, double *e)
{
    zc om = w_re + w_im * I;
    zc al = alpha[0] + alpha[1] * I, be = beta[0] + beta[1] * I;
    if (ipars == 0)
        _QMrfmodulePcal_e_inv(&om, &ray_p, &al, &be, &rho, (zc *)e);
    else
        _QMrfmodulePcal_e_inv_par(&om, &ray_p, &al, &be, &rho, (zc *)e, &ipars);
}

int refprobe_nextpow2(int n)
{
    int out = 0;
    nextpow2_(&n, &out);
    return out;
}

/* deconit.f90:15-32: gauss[nt/2 + 1] = exp(-0.25 (2 pi_f32 f / f0)^2), f = i / (nt dt) */
void refprobe_gauss_filter(int nt, double dt, double f0, double *gauss)
{
    gauss_filter_(&nt, &dt, &f0, gauss);
}
