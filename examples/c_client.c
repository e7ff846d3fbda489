/* Plain C client of the C ABI (include/rfsurf.h): what a cgo / JNI / Fortran-2003 binding would do.
 *
 *   gcc -std=c99 -I.. c_client.c -o c_client -L../rfsurfhmc_amd -l:librfsurf_hip.so -Wl,-rpath,$PWD/../rfsurfhmc_amd
 *
 * Computes the Rayleigh phase velocities and the vs kernel of the reference's param.yaml model
 * (libsurf.forward / libsurf.adjoint_kernel, src/SWD/main.cpp:14-82) and its P receiver function
 * (librf.forward, src/RF/main.cpp:17-62, method "freq"), and prints them. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "include/rfsurf.h"

#define NL 7
#define NP 6
#define NT 125

static void empirical(const double* vs, double* vp, double* rho)
{   /* model/model_surf.py:47-79 */
    for (int i = 0; i < NL; i++) {
        double b = vs[i];
        double a = 0.9409 + 2.0947 * b - 0.8206 * b * b + 0.2683 * b * b * b - 0.0251 * b * b * b * b;
        vp[i] = a;
        rho[i] = 1.6612 * a - 0.4721 * a * a + 0.0671 * a * a * a - 0.0043 * a * a * a * a + 0.000106 * a * a * a * a * a;
    }
}

int main(void)
{
    const double thk[NL] = {6., 6., 13., 5., 10., 30., 0.}, vs[NL] = {3.2, 2.8, 3.46, 3.3, 3.9, 4.5, 4.7};
    const double period[NP] = {5., 6., 7., 8., 9., 10.};
    double vp[NL], rho[NL], qa[NL], qb[NL];
    double c[NP], ka[NP * NL], kb[NP * NL], kr[NP * NL], kh[NP * NL], rf[NT];
    int32_t flag = 0;
    rfs_ctx* ctx = NULL;
    empirical(vs, vp, rho);
    for (int i = 0; i < NL; i++) qa[i] = qb[i] = 9999.0;
    if (rfs_create(&ctx, 0, 16, 32) != RFS_OK) { fprintf(stderr, "no usable MI355X (there is no CPU fallback)\n"); return 2; }
    if (rfs_swd_kernel(ctx, 1, NL, thk, vp, vs, rho, NP, period, RFS_WAVE_RC, 0, 0, c, ka, kb, kr, kh, &flag) != RFS_OK) {
        fprintf(stderr, "%s\n", rfs_last_error(ctx)); return 1;
    }
    printf("flag %d\nc", (int)flag);
    for (int k = 0; k < NP; k++) printf(" %.12f", c[k]);
    printf("\ndcdb[T=5s]");
    for (int i = 0; i < NL; i++) printf(" %.12e", kb[i]);
    rfs_rf_params par = {0.045, NT, 0.4, 1.5, 5.0, 0.001, RFS_RF_P, RFS_RF_FREQ};
    if (rfs_rf_forward(ctx, 1, NL, thk, rho, vp, vs, qa, qb, &par, rf) != RFS_OK) {
        fprintf(stderr, "%s\n", rfs_last_error(ctx)); return 1;
    }
    printf("\nrf[8:16]");
    for (int t = 8; t < 16; t++) printf(" %.12e", rf[t]);
    printf("\n");
    rfs_destroy(ctx);
    return 0;
}
