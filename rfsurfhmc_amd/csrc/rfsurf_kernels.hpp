// HIP kernels (gfx950, wave64) for the RF + SWD misfit/gradient hot path.
// Lane-level math lives in rf_math.hpp / swd_math.hpp; the kernels below only map
// (chain, frequency | period) work items onto wavefronts, move data and reduce.
//
// Data layout in HBM (all f64 unless noted; nchain = chains of the current call):
//   x        [chain][2n]                 model vectors (vs, thk)                     (input)
//   lc       [chain][n] RfLayer          frequency-independent RF layer constants
//   cr       [chain][2][n]               chain-rule factors dadb, drda*dadb
//   mdl      [4][n][chain] f32           SWD model (thk, vp, vs, rho) rounded to float32,
//                                        chain-minor so that "lane = chain" loads coalesce
//   RR       [chain][4][n2p]             Re/Im R21, Re/Im R22 per frequency
//   Rs       [chain][n-1][8][n2p]        pass-A row vectors r_j (scratch, frequency-minor)
//   spec     [chain][n2] complex         RF spectrum -> rocFFT c2r -> tser [chain][nft]
//   wres     [chain][nft]                weighted residual -> rocFFT r2c -> W [chain][n2] complex
//   PG       [chain][npart][4][n]        per-wave partial gradient sums (deterministic reduce)
//   croot    [seq][nper][chain]          phase velocities per root-search sequence
//   cds      [n][6][item]                compound up-sweep scratch of the eigenfunction pass
//   krn      [item-class][4][n][chain]   kernels before their per-item scales (swd_krn): slots 0-1 = d(c)/d(vs), interface partial
//                                        (chain-ruled storage: joint evaluation, warm start), or the four raw classes
//                                        d(c)/d(alpha,beta,rho,interface) (B1)
#pragma once
#include <hip/hip_runtime.h>
#include "rf_math.hpp"
#include "swd_math.hpp"

namespace rfs {

constexpr int MAXL = 128;   // layers per model (two register slots in the pass-B reduction)

struct RfFreq {             // frequency axis + RF scalars shared by the RF kernels
    double dt, sigma, p, f0, t0, water;
    int nft, n2, n2p, nt, rf_type, fwd_order;
    int method, pi64;       // method: RFS_RF_* ; pi64: f64 pi on the frequency axis (cal_rf_par_time_all only)
    // Band limit of the adjoint (fused gradient only): frequencies k >= nk carry a Gaussian weight exp(-(w/2f0)^2) below
    // eps * water and cannot reach the gradient in double precision; pass A still sweeps them (the water level is a
    // maximum over ALL frequencies, RFModule.f90:396-398) but keeps no rows for them, pass B does not run them.
    // nk = n2, nkp = n2p: no limit (B1 kernel_all, time-domain method).  nkp = row-scratch stride (nk padded to 16).
    int nk, nkp;
    double peel_emax;       // growth exponent up to which a chain's rows are rebuilt by peeling (rf_growth_exponent; 0 = never)
    double e32max;          // growth exponent (at the Nyquist frequency) up to which pass A sweeps a chain's frequencies beyond
                            // the band in float32 (rf_row_step_f32, rf_f32_emax; 0 = never)
};

__device__ __forceinline__ double rf_wk(const RfFreq& f, int k) {
    // RFModule.f90:384  w = 1/nft/dt*(it-1)*2*pi   (cal_rf_freq :231 and cal_rf_time :173 use 1/dt/nft order);
    // pi = atan(1.0)*4.0 in default real everywhere except cal_rf_par_time_all (:96 atan(1.0_dp)*4.0_dp)
    const double pi = f.pi64 ? 3.14159265358979323846 : RF_PI32;
    if (f.fwd_order) return (1.0 / f.dt / f.nft) * k * 2 * pi;
    return 1.0 / f.nft / f.dt * k * 2.0 * pi;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// Wave-wide sum on the VALU (DPP moves of the two 32-bit halves + v_add_f64): no LDS-crossbar (ds_bpermute) traffic
// and a short dependent chain, which matters in the sweeps that reduce four values per layer at low occupancy.
// The total is returned wave-uniform (read from lane 63).  Fixed summation order -> deterministic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_step(double v) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_uniform(double v) {
    v = dpp_add_step<0xb1, 0xf>(v);      // quad_perm [1,0,3,2]: pair sums
    v = dpp_add_step<0x4e, 0xf>(v);      // quad_perm [2,3,0,1]: quad sums (in all four lanes)
    v = dpp_add_step<0x114, 0xf>(v);     // row_shr:4  (zero fill)
    v = dpp_add_step<0x118, 0xf>(v);     // row_shr:8  -> row totals in lanes 12..15 of each row
    v = dpp_add_step<0x142, 0xa>(v);     // row_bcast:15 into rows 1, 3
    v = dpp_add_step<0x143, 0xc>(v);     // row_bcast:31 into rows 2, 3 -> lane 63 holds the wave total
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63),
                            __builtin_amdgcn_readlane(__double2loint(v), 63));
}
// Four wave-wide sums at once: two transposing butterfly steps inside each quad leave every lane with ONE partial sum
// (lane l of a quad holds value (l & 3)), so that only one value per lane instead of four goes through the four
// cross-quad steps.  Returns the total of value q in out[q], wave-uniform.  About half the instructions of the
// four separate sums of a pass-B layer; fixed order -> deterministic.
template <int CTRL>
__device__ __forceinline__ double dpp_xchg(double v) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void wave_sum4_uniform(const double v[4], double out[4]) {
    const int lane = threadIdx.x & 63;
    const bool b0 = lane & 1, b1 = lane & 2;
    // step 1 (partner = lane ^ 1): even lanes collect values 0, 1; odd lanes values 2, 3
    double ka = b0 ? v[2] : v[0], kb = b0 ? v[3] : v[1];
    double sa = b0 ? v[0] : v[2], sb = b0 ? v[1] : v[3];
    ka += dpp_xchg<0xb1>(sa); kb += dpp_xchg<0xb1>(sb);          // quad_perm [1,0,3,2]
    // step 2 (partner = lane ^ 2): lanes 0,1 of a quad keep the first of their pair, lanes 2,3 the second
    double k = b1 ? kb : ka, s2 = b1 ? ka : kb;
    k += dpp_xchg<0x4e>(s2);                                      // quad_perm [2,3,0,1]
    // lane l of every quad now holds the quad's sum of value q(l): l&3 = 0 -> v0, 1 -> v2, 2 -> v1, 3 -> v3
    k = dpp_add_step<0x114, 0xf>(k);     // row_shr:4
    k = dpp_add_step<0x118, 0xf>(k);     // row_shr:8  -> lanes 12..15 of each row: the row's four totals
    // across the four rows the lanes must keep their identity (row_bcast would spread lane 15 only): two shuffles
    k += __shfl_xor(k, 16, 64);
    k += __shfl_xor(k, 32, 64);
    const int lo = __double2loint(k), hi = __double2hiint(k);
    out[0] = __hiloint2double(__builtin_amdgcn_readlane(hi, 12), __builtin_amdgcn_readlane(lo, 12));
    out[2] = __hiloint2double(__builtin_amdgcn_readlane(hi, 13), __builtin_amdgcn_readlane(lo, 13));
    out[1] = __hiloint2double(__builtin_amdgcn_readlane(hi, 14), __builtin_amdgcn_readlane(lo, 14));
    out[3] = __hiloint2double(__builtin_amdgcn_readlane(hi, 15), __builtin_amdgcn_readlane(lo, 15));
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

// rf_growth_exponent with the layers spread over the lanes of the wavefront (every lane gets the sum): ~20 instructions
// instead of a loop over the layers in every lane
__device__ __forceinline__ double rf_growth_exponent_wave(const RfLayer* __restrict__ L, int n, double sigma, double wmax) {
    double e = 0.0;
    for (int j = threadIdx.x & 63; j < n - 1; j += 64)
        e += L[j].h * (sigma * (fabs(L[j].pvb.im) + fabs(L[j].pva.im)) + wmax * (fabs(L[j].pva.re) + fabs(L[j].pvb.re)));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) e += __shfl_xor(e, off, 64);
    return e;
}

// f64 per-layer constants of the split secular function, from the float32-rounded model:
// mdlc[(m*6 + q)*nchain + chain], q = d, 1/alpha, 1/beta, beta, rho, 1/rho
__device__ __forceinline__ void swd_store_layerc(double* __restrict__ mdlc, int m, int chain, int nchain,
                                                 float d, float a, float b, float rho)
{
    double* o = mdlc + (size_t)m * 6 * nchain + chain;
    o[0] = (double)d; o[(size_t)nchain] = 1.0 / (double)a; o[(size_t)2 * nchain] = 1.0 / (double)b;
    o[(size_t)3 * nchain] = (double)b; o[(size_t)4 * nchain] = (double)rho; o[(size_t)5 * nchain] = 1.0 / (double)rho;
}

// ---------------------------------------------------------------------------------------
// K_PREP: one thread per (chain, layer).  model_rf.py:52-77 / model_surf.py:47-79 empirical
// relations, RfLayer constants, float32 SWD model.
// ---------------------------------------------------------------------------------------
// The drift of a leapfrog step (pyhmc/hmc.py:164-183: x += dt M^-1 p with mirror reflection at the bounds, preceded by
// the half kick a deferred start left open), one component.  Runs inside k_prep_joint for the flow entries,
// whose thread (chain, layer) owns the components vs_j and thk_j.  p == nullptr: off.
// hmc.py:121-137 reflects until the point is inside its bounds, however far out it was.  64 reflections are made as the reference
// makes them; a point still outside then -- a chain whose momentum has blown up (a gradient next to the reference's not-a-number
// kernels): x of 1e4 .. 1e300 -- is folded in closed form, the same point up to rounding, instead of being evaluated where it is:
// a model with vs = 6e4 km/s costs the reference-semantics search a scan of 1.3e7 cells, 31 s during which every other chain of
// the batch waits (round 6: one chain of 56 of configs[0]'s sampler, step 42).  Not a number / infinite: the middle of the
// bounds -- the momentum stays what it is, so the trajectory's energy is not finite and the trajectory is rejected.
__device__ __forceinline__ void flow_mirror(double& xv, double& pv, double lo, double hi) {
    for (int it = 0; it < 64 && (xv > hi || xv < lo); it++) {
        if (xv > hi) { xv = 2 * hi - xv; pv = -pv; }
        if (xv < lo) { xv = 2 * lo - xv; pv = -pv; }
    }
    if (xv >= lo && xv <= hi) return;
    const double w = hi - lo;
    if (!(w > 0.0) || !(fabs(xv) < 1.0e300)) { xv = lo + 0.5 * w; return; }
    double y = fmod(xv - lo, 2.0 * w);
    if (y < 0.0) y += 2.0 * w;
    if (y > w) { y = 2.0 * w - y; pv = -pv; }
    xv = lo + y;
}
struct FlowPre {
    const double* minv; const double* dt; const int* rem; const int* fresh; const int* ok; const double* bounds;
    double* x; double* p; const double* gsave; const int* kick; int* wforce;
    const int* pend;        // [chain] 1: handed back to the full search in the step before and left out of it (rfs_set_option flow_async_handback): its drift has been made
};
__device__ __forceinline__ void flow_drift(const FlowPre& F, int chain, int i, int nx) {
    // option swd_exact_final: start and end models of a trajectory by the reference-semantics search (k_swd_warm's force)
    if (F.wforce && i == 0) F.wforce[chain] = F.fresh[chain] || F.rem[chain] == 1;
    if (F.fresh[chain] || F.rem[chain] <= 0 || !F.ok[chain]) return;
    if (F.pend && F.pend[chain]) return;                       // this chain's evaluation of an earlier step is only completed now (or still waits for its search)
    const size_t g = (size_t)chain * nx + i;
    double pv = F.p[g];
    if (F.kick && F.kick[chain]) pv = pv - F.dt[chain] * F.gsave[g] * 0.5;      // the half kick a deferred start left open (hmc.py:164)
    // (same expression as in k_flow_post's start branch: the two forms give the same p bit for bit)
    double xv = F.x[g] + F.dt[chain] * (pv * (F.minv ? F.minv[i] : 1.0));
    flow_mirror(xv, pv, F.bounds[2 * i], F.bounds[2 * i + 1]);
    F.x[g] = xv; F.p[g] = pv;
}

__global__ void k_prep_joint(int nchain, int n, const double* x /* may be fpre.x */, int has_rf, double ray_p,
                             RfLayer* __restrict__ lc, double* __restrict__ cr, int has_swd,
                             float* __restrict__ mdl, double* __restrict__ mdlc,
                             double* __restrict__ zero_d, size_t nzero_d, int* __restrict__ zero_i, size_t nzero_i,
                             double* __restrict__ xw, double* __restrict__ dxT, double* __restrict__ crT, FlowPre fpre,
                             int* __restrict__ zero_c1 = nullptr, int* __restrict__ zero_c2 = nullptr)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    // (the list counters of the warm search's rounds and of the reference-root stage: a hipMemsetAsync of 32 bytes in front of
    // those kernels is a fill launch of 40-80 us on the step's critical chain)
    if (g < 8) { if (zero_c1) zero_c1[g] = 0; if (zero_c2 && g < 4) zero_c2[g] = 0; }
    // per-step clearing for the early eigenfunction launch (roots: zero = not final; done map), folded in here
    // instead of two memset launches in front of the fork
    for (size_t i = g; i < nzero_d; i += (size_t)gridDim.x * blockDim.x) zero_d[i] = 0.0;
    for (size_t i = g; i < nzero_i; i += (size_t)gridDim.x * blockDim.x) zero_i[i] = 0;
    if (g >= nchain * n) return;
    int chain = g / n, j = g - chain * n;
    if (fpre.p) { flow_drift(fpre, chain, j, 2 * n); flow_drift(fpre, chain, n + j, 2 * n); }      // (x below is fpre.x)
    double vs = x[(size_t)chain * 2 * n + j], thk = x[(size_t)chain * 2 * n + n + j];
    double vp = 0.9409 + 2.0947 * vs - 0.8206 * (vs * vs) + 0.2683 * (vs * vs * vs) - 0.0251 * (vs * vs * vs * vs);
    double vp2 = vp * vp;
    double rho = 1.6612 * vp - 0.4721 * vp2 + 0.0671 * (vp2 * vp) - 0.0043 * (vp2 * vp2) + 0.000106 * (vp2 * vp2 * vp);
    double drda = 1.6612 - 0.4721 * 2 * vp + 0.0671 * 3 * vp2 - 0.0043 * 4 * (vp2 * vp) + 0.000106 * 5 * (vp2 * vp2);
    double dadb = 2.0947 - 0.8206 * 2 * vs + 0.2683 * 3 * (vs * vs) - 0.0251 * 4 * (vs * vs * vs);
    cr[((size_t)chain * 2 + 0) * n + j] = dadb;
    cr[((size_t)chain * 2 + 1) * n + j] = drda * dadb;
    if (xw) {
        // warm start of the root search (k_swd_warm): the model's change since the evaluation before this one and the
        // chain-rule factors, chain-minor; xw then becomes this model
        double* w = xw + (size_t)chain * 2 * n;
        dxT[(size_t)j * nchain + chain] = vs - w[j];
        dxT[(size_t)(n + j) * nchain + chain] = thk - w[n + j];
        w[j] = vs; w[n + j] = thk;
    }
    if (crT) {     // chain-rule factors, chain-minor: the eigenfunction pass stores d c / d vs with them (k_swd_eigen)
        crT[(size_t)j * nchain + chain] = dadb;
        crT[(size_t)(n + j) * nchain + chain] = drda * dadb;
    }
    if (has_rf) rf_make_layer(lc[(size_t)chain * n + j], thk, rho, vp, vs, 9999.0, 9999.0, ray_p);
    if (has_swd) {
        size_t s = (size_t)n * nchain;
        mdl[0 * s + (size_t)j * nchain + chain] = (float)thk;
        mdl[1 * s + (size_t)j * nchain + chain] = (float)vp;
        mdl[2 * s + (size_t)j * nchain + chain] = (float)vs;
        mdl[3 * s + (size_t)j * nchain + chain] = (float)rho;
        swd_store_layerc(mdlc, j, chain, nchain, (float)thk, (float)vp, (float)vs, (float)rho);
    }
}

// B1 variants: explicit thk, rho, vp, vs (, qa, qb) arrays [chain][n]
__global__ void k_prep_rf_b1(int nchain, int n, const double* thk, const double* rho, const double* vp,
                             const double* vs, const double* qa, const double* qb, double ray_p,
                             RfLayer* __restrict__ lc)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nchain * n) return;
    rf_make_layer(lc[g], thk[g], rho[g], vp[g], vs[g], qa[g], qb[g], ray_p);
}

__global__ void k_prep_swd_b1(int nchain, int n, const double* thk, const double* vp, const double* vs,
                              const double* rho, float* __restrict__ mdl, double* __restrict__ mdlc)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nchain * n) return;
    int chain = g / n, j = g - chain * n;
    size_t s = (size_t)n * nchain, o = (size_t)j * nchain + chain;
    mdl[0 * s + o] = (float)thk[g]; mdl[1 * s + o] = (float)vp[g];
    mdl[2 * s + o] = (float)vs[g];  mdl[3 * s + o] = (float)rho[g];
    swd_store_layerc(mdlc, j, chain, nchain, (float)thk[g], (float)vp[g], (float)vs[g], (float)rho[g]);
}

// Per-family SWD models, one thread per chain (sequential over the layers: the flattening accumulates depth).
//   sphere && wantR : mdlSR = float32 flattened Rayleigh model for the root search (+ its f64 layer constants in
//                     mdlc, replacing the flat ones), sphR = [7][n][chain] f64 bldsph arrays zd,za,zb,zrho,vtp,dtp,rtp
//   wantL           : mdlL = [5][n][chain] float32 Love search model d,a,b,rho,a' (flat copy or flattened), mdlcL its
//                     f64 layer constants;
//                     a' = P velocity 1.732 vs of _LoveGroup (surfdisp.cpp:132); sphere: sphL like sphR
__global__ void k_prep_swd_family(int nchain, int n, const float* __restrict__ mdl, int sphere, int wantR, int wantL,
                                  float* __restrict__ mdlSR, double* __restrict__ mdlc, double* __restrict__ sphR,
                                  float* __restrict__ mdlL, double* __restrict__ sphL, double* __restrict__ mdlcL)
{
    int chain = blockIdx.x * blockDim.x + threadIdx.x;
    if (chain >= nchain) return;
    const size_t s = (size_t)n * nchain;
    const long st = nchain;
    const float *d = mdl + chain, *a = mdl + s + chain, *b = mdl + 2 * s + chain, *r = mdl + 3 * s + chain;
    if (sphere && wantR) {
        float *od = mdlSR + chain, *oa = mdlSR + s + chain, *ob = mdlSR + 2 * s + chain, *orr = mdlSR + 3 * s + chain;
        swd_flatten_f32(false, n, d, a, b, r, st, od, oa, ob, orr, st);
        for (int m = 0; m < n; m++) swd_store_layerc(mdlc, m, chain, nchain, od[m * st], oa[m * st], ob[m * st], orr[m * st]);
        double* z = sphR + chain;
        swd_bldsph(false, n, d, a, b, r, st, z, z + s, z + 2 * s, z + 3 * s, z + 4 * s, z + 5 * s, z + 6 * s, st);
    }
    if (wantL) {
        float *od = mdlL + chain, *oa = mdlL + s + chain, *ob = mdlL + 2 * s + chain, *orr = mdlL + 3 * s + chain,
              *oa2 = mdlL + 4 * s + chain;
        for (int m = 0; m < n; m++) oa2[m * st] = (float)(1.732 * b[m * st]);
        if (sphere) {
            swd_flatten_f32(true, n, d, a, b, r, st, od, oa, ob, orr, st);
            const double ar = 6370.0;                       // a' goes through the same velocity mapping as a
            double dr = 0.0, r0 = ar;
            for (int m = 0; m < n; m++) {
                dr = dr + (double)((m == n - 1) ? 1.0f : d[m * st]);
                double r1 = ar - dr, tmp = (ar + ar) / (r0 + r1);
                oa2[m * st] = (float)((double)oa2[m * st] * tmp);
                r0 = r1;
            }
            double* z = sphL + chain;
            swd_bldsph(true, n, d, a, b, r, st, z, z + s, z + 2 * s, z + 3 * s, z + 4 * s, z + 5 * s, z + 6 * s, st);
        } else {
            for (int m = 0; m < n; m++) { od[m * st] = d[m * st]; oa[m * st] = a[m * st]; ob[m * st] = b[m * st]; orr[m * st] = r[m * st]; }
        }
        // f64 layer constants of the Love search model (the lanes-per-item search reads thickness, 1/beta, beta, rho, 1/rho)
        for (int m = 0; m < n; m++) swd_store_layerc(mdlcL, m, chain, nchain, od[m * st], oa[m * st], ob[m * st], orr[m * st]);
    }
}

// ---------------------------------------------------------------------------------------
// K1 pass A: lane = frequency (TAIL=false: block = one chain's 64..256 frequencies, layer
// constants are wave-uniform -> scalar loads) or lane = chain at the last frequency
// (TAIL=true: n2 = nft/2 + 1 is odd, the Nyquist bin is swept "chain-wide" instead).
// ---------------------------------------------------------------------------------------
// one frequency's sweep up the stack in f64: R21, R22 (NaN scrubbed, RFModule.f90:662-667)
__device__ __forceinline__ void rf_r21_r22(const RfFreq& f, const V4& r, cplx& r21, cplx& r22) {
    const int c21 = (f.rf_type == 1) ? 0 : 1, c22 = 1 - c21;
    r21 = r.v[c21];
    r22 = (f.rf_type == 1) ? mul_i(r.v[c22]) : -mul_i(r.v[c22]);
    if (r21.re != r21.re || r21.im != r21.im) r21 = C(0.0);
    if (r22.re != r22.re || r22.im != r22.im) r22 = C(0.0);
}
__device__ __forceinline__ void rf_sweep_plain(const RfLayer* __restrict__ L, int n, const RfFreq& f, int k, cplx& r21, cplx& r22) {
    const cplx omega = C(rf_wk(f, k), -f.sigma);
    V4 r = rf_einv_row(L[n - 1], f.rf_type);
    for (int j = n - 2; j >= 0; j--) {
        RfHyp H; RfA A;
        rf_hyp(L[j], omega, H);
        rf_build_A(L[j], H, A);
        r = rf_row_times_A(r, A);
    }
    rf_r21_r22(f, r, r21, r22);
}

// TAIL: the Nyquist bin, lane = chain; otherwise lane = frequency k of chunk `cy` of ONE chain (block-uniform: its layer
// constants come through scalar loads).  Both are rows of one grid (k_rf_passA below).
template <bool TAIL>
__device__ __forceinline__ void
rf_passA_rows(int nchain, int n, const RfFreq& f, const RfLayer* __restrict__ lc, double* __restrict__ RR,
              double* __restrict__ Rs, double* __restrict__ RT, int* __restrict__ slist, int* __restrict__ scount,
              int* __restrict__ scount_next, int* __restrict__ hi32, const int cy, double* __restrict__ Hs = nullptr)
{
    int chain, k;
    bool live = true;
    if (TAIL) {
        chain = blockIdx.x * blockDim.x + threadIdx.x; k = f.n2 - 1;
        if (chain >= nchain) return;
    } else {
        // grid = (chain, chunk of frequencies): workgroups go round-robin over the 8 XCDs by their linear index, so the chain
        // must be the fast index -- with the chunk there, each XCD would get ONE kind of chunk (all f64 band chunks on one XCD)
        chain = blockIdx.x; k = cy * blockDim.x + threadIdx.x;
        if (k >= f.n2 - 1) { live = false; k = f.n2 - 2; }      // (kept until the wave-wide sums below are done)
    }
    const RfLayer* L = lc + (size_t)chain * n;
    cplx omega = C(rf_wk(f, k), -f.sigma);
    const size_t n2p = f.n2p, nkp = f.nkp;
    // RT given = row peeling allowed: rows are then stored only for a chain whose layer matrices grow too much to be
    // peeled off again (rf_growth_exponent; the same test picks the path in pass B)
    const bool store = Rs && (!RT || (TAIL ? rf_growth_exponent(L, n, f.sigma, rf_wk(f, f.nk))
                                           : rf_growth_exponent_wave(L, n, f.sigma, rf_wk(f, f.nk))) > f.peel_emax);
    // beyond the band: float32 for the chains whose matrices stay tame up to the Nyquist frequency (k_rf_mid1 takes it from there)
    bool c32 = false;
    if (!TAIL && f.e32max > 0.0 && f.nk < f.n2) {
        c32 = rf_growth_exponent_wave(L, n, f.sigma, rf_wk(f, f.n2 - 1)) <= f.e32max;
        if (hi32 && cy == 0 && threadIdx.x == 0) hi32[chain] = c32 ? 1 : 0;
    }
    // (the chains that keep stored rows, for pass B's launch over them: normally none)
    if (!TAIL && RT && store && slist && cy == 0 && threadIdx.x == 0) slist[atomicAdd(scount, 1)] = chain;
    if (!TAIL && scount_next && blockIdx.x == 0 && cy == 0 && threadIdx.x == 0) *scount_next = 0;   // the NEXT evaluation's counter
    if (!live) return;
    double* o = RR + (size_t)chain * 4 * n2p + k;
    if (!TAIL && c32 && k - (int)(threadIdx.x & 63) >= f.nk) {          // a whole wavefront beyond the band
        const V4 r0 = rf_einv_row(L[n - 1], f.rf_type);
        V4f r;
#pragma unroll
        for (int i = 0; i < 4; i++) r.v[i] = to_f32(r0.v[i]);
        for (int j = n - 2; j >= 0; j--) r = rf_row_step_f32(L[j], omega, r);
        V4 rd;
#pragma unroll
        for (int i = 0; i < 4; i++) rd.v[i] = C((double)cf_re(r.v[i]), (double)cf_im(r.v[i]));
        cplx r21, r22;
        rf_r21_r22(f, rd, r21, r22);
        // (rf_r21_r22 scrubs NaN to 0 like the reference, RFModule.f90:662-667 -- for a FLOAT32 result that would understate the
        // maxima k_rf_mid1 builds its proof on: an overflowed or undefined float32 value goes on as +inf instead, the
        // verdict is then "sweep again in f64" and the f64 sweep decides what the reference's scrub sees)
        bool finite32 = true;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float a = cf_re(r.v[i]), b = cf_im(r.v[i]);
            finite32 = finite32 && (a - a == 0.0f) && (b - b == 0.0f);
        }
        if (!finite32) r21 = C(__longlong_as_double(0x7ff0000000000000LL), 0.0);
        o[0] = r21.re; o[n2p] = r21.im; o[2 * n2p] = r22.re; o[3 * n2p] = r22.im;
        return;
    }
    V4 r = rf_einv_row(L[n - 1], f.rf_type);
    double* rs = (store && k < f.nk) ? Rs + ((size_t)chain * (n - 1)) * 8 * nkp + k : nullptr;
    // the transcendental numbers of every (layer, band frequency) for pass B (rf_hyp_base; [chain][layer][6][nkp])
    double* hs = (!TAIL && Hs && !store && k < f.nk) ? Hs + ((size_t)chain * (n - 1)) * 6 * nkp + k : nullptr;
    for (int j = n - 2; j >= 0; j--) {
        if (rs) {
            double* q = rs + (size_t)j * 8 * nkp;
#pragma unroll
            for (int i = 0; i < 4; i++) { q[(2 * i) * nkp] = r.v[i].re; q[(2 * i + 1) * nkp] = r.v[i].im; }
        }
        RfHyp H; RfA A; RfHypB B;
        rf_hyp_base(L[j], omega, B);
        if (hs) {
            double* q = hs + (size_t)j * 6 * nkp;
            q[0] = B.e1; q[nkp] = B.c1; q[2 * nkp] = B.s1; q[3 * nkp] = B.e2; q[4 * nkp] = B.c2; q[5 * nkp] = B.s2;
        }
        rf_hyp_from(L[j], omega, B, H);
        rf_build_A(L[j], H, A);
        r = rf_row_times_A(r, A);
    }
    if (RT && !store && k < f.nk) {  // pass B peels the layers off this FINAL row itself (rf_row_times_Ainv): no row scratch
        double* q = RT + (size_t)chain * 8 * nkp + k;
#pragma unroll
        for (int i = 0; i < 4; i++) { q[(2 * i) * nkp] = r.v[i].re; q[(2 * i + 1) * nkp] = r.v[i].im; }
    }
    cplx r21, r22;
    rf_r21_r22(f, r, r21, r22);
    o[0] = r21.re; o[n2p] = r21.im; o[2 * n2p] = r22.re; o[3 * n2p] = r22.im;
}
// One launch for all frequencies: grid = (chain, 1 + chunks).  Row 0 holds the Nyquist bin (lane = chain: its first
// ceil(nchain / blockDim) blocks work, the rest leave at once) -- dispatched FIRST, so its one-wavefront-per-64-chains sweep
// runs beside the bulk instead of as a launch of its own behind it (0.11 ms alone, 0.3-0.4 ms in the shared step:
// profiles/r04_step_timeline.txt); rows 1 .. chunks: chunk y - 1 of chain x.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5)))      // (the bulk rows need 85 VGPRs; the Nyquist row's per-lane layer constants would cost the whole kernel a wavefront per SIMD)
k_rf_passA(int nchain, int n, RfFreq f, const RfLayer* __restrict__ lc, double* __restrict__ RR,
           double* __restrict__ Rs, double* __restrict__ RT, int* __restrict__ slist, int* __restrict__ scount,
           int* __restrict__ scount_next, int* __restrict__ hi32, double* __restrict__ Hs)
{
    if (blockIdx.y == 0) {                       // (block-uniform)
        if ((size_t)blockIdx.x * blockDim.x >= (size_t)nchain) return;
        rf_passA_rows<true>(nchain, n, f, lc, RR, Rs, RT, slist, scount, nullptr, nullptr, 0);
    } else {
        rf_passA_rows<false>(nchain, n, f, lc, RR, Rs, RT, slist, scount, scount_next, hi32, (int)blockIdx.y - 1, Hs);
    }
}
// ---------------------------------------------------------------------------------------
// K2 mid 1: per chain -- water level (max over all frequencies, RFModule.f90:396-398,
// 411-413) and the RF spectrum S = conj(R21) R22 G e^{-i w t0} / fai (:401), with the
// imaginary parts of DC / Nyquist zeroed (FFTW's c2r ignores them, rocFFT must not see them).
// ---------------------------------------------------------------------------------------
// the first half of k_rf_mid1 (water-level maxima; float32 verdict and re-sweep), shared by the fused kernel
__device__ __forceinline__ void rf_mid_maxima(int n, const RfFreq& f, const RfLayer* __restrict__ lc, double* __restrict__ rr, int chain,
                                              const int* __restrict__ hi32, unsigned long long* __restrict__ stat32,
                                              double (*red)[4], double& m1, double& m2)
{
    const int tid = threadIdx.x, wv = tid >> 6, nw = blockDim.x >> 6;
    bool h32 = hi32 && hi32[chain] != 0;
    m1 = 0.0; m2 = 0.0;
    for (int pass = 0; pass < 2; pass++) {
        double b1 = 0.0, b2 = 0.0, h1 = 0.0, h2 = 0.0, lo1 = 1.0e300, lo2 = 1.0e300;
        for (int k = tid; k < f.n2; k += blockDim.x) {
            cplx r21 = C(rr[k], rr[f.n2p + k]);
            double wa = (r21 * conj(r21)).re;
            cplx sq = r21 * r21;
            double wb = (sq * conj(sq)).re;
            if (h32 && k >= f.nk) { h1 = fmax(h1, wa); h2 = fmax(h2, wb); }
            else { b1 = fmax(b1, wa); b2 = fmax(b2, wb); lo1 = fmin(lo1, wa); lo2 = fmin(lo2, wb); }
        }
        b1 = wave_max(b1); b2 = wave_max(b2); h1 = wave_max(h1); h2 = wave_max(h2); lo1 = -wave_max(-lo1); lo2 = -wave_max(-lo2);
        __syncthreads();
        if ((tid & 63) == 0) { red[0][wv] = b1; red[1][wv] = b2; red[2][wv] = h1; red[3][wv] = h2; red[4][wv] = lo1; red[5][wv] = lo2; }
        __syncthreads();
        for (int i = 0; i < nw; i++) {
            b1 = fmax(b1, red[0][i]); b2 = fmax(b2, red[1][i]); h1 = fmax(h1, red[2][i]); h2 = fmax(h2, red[3][i]);
            lo1 = fmin(lo1, red[4][i]); lo2 = fmin(lo2, red[5][i]);
        }
        m1 = fmax(b1, h1); m2 = fmax(b2, h2);
        if (!h32) break;
        const int verdict = rf_f32_decide(f.water, b1, b2, h1, h2, lo1, lo2);
        if (verdict == 0) { m1 = b1; m2 = b2; }
        if (tid == 0 && stat32 && pass == 0) atomicAdd(&stat32[2 + (chain & 63)], 1ull);      // (64 slots: one address would serialise)
        if (verdict != 2) break;
        // the rare chain: its frequencies beyond the band again, in f64 (the Nyquist bin always is)
        if (tid == 0 && stat32) atomicAdd(&stat32[0], 1ull);
        const RfLayer* L = lc + (size_t)chain * n;
        for (int k = f.nk + tid; k < f.n2 - 1; k += blockDim.x) {
            cplx r21, r22;
            rf_sweep_plain(L, n, f, k, r21, r22);
            rr[k] = r21.re; rr[f.n2p + k] = r21.im; rr[2 * f.n2p + k] = r22.re; rr[3 * f.n2p + k] = r22.im;
        }
        __syncthreads();
        h32 = false;
    }
}

__global__ void __launch_bounds__(256)
k_rf_mid1(int n, RfFreq f, const RfLayer* __restrict__ lc, double* __restrict__ RR, double* __restrict__ wmax2,
          cplx* __restrict__ spec, const int* __restrict__ hi32, unsigned long long* __restrict__ stat32)
{
    __shared__ double red[6][4];
    int chain = blockIdx.x, tid = threadIdx.x;
    double* rr = RR + (size_t)chain * 4 * f.n2p;
    // Pass A swept this chain's frequencies beyond the band in float32 (hi32): the maxima are taken apart -- band values are
    // exact, the others carry a relative error below RF_F32_MARGIN.  Where the band holds both maxima, or no band
    // frequency can reach a water level set by (1 + margin) x the others' maxima, every number that reaches the results
    // with a weight above exp(-(w_nk/2f0)^2) is what the all-f64 sweep gives; otherwise the block sweeps those
    // frequencies again in f64 (statistic rf_f32_resweeps) and nothing of the float32 pass is left (rf_mid_maxima).
    double m1, m2;
    rf_mid_maxima(n, f, lc, rr, chain, hi32, stat32, red, m1, m2);
    if (tid == 0) wmax2[chain] = m2;
    for (int k = tid; k < f.n2; k += blockDim.x) {
        cplx r21 = C(rr[k], rr[f.n2p + k]), r22 = C(rr[2 * f.n2p + k], rr[3 * f.n2p + k]);
        double w = rf_wk(f, k);
        double wa = (r21 * conj(r21)).re;
        double fai = fmax(wa, f.water * m1);
        double g = exp(-((w / 2 / f.f0) * (w / 2 / f.f0)));
        double s, c; sincos(w * f.t0, &s, &c);
        cplx S = (conj(r21) * r22) * (g / fai) * C(c, -s);
        if (k == 0 || k == f.n2 - 1) S.im = 0.0;
        spec[(size_t)chain * f.n2 + k] = S;
    }
}

// ---------------------------------------------------------------------------------------
// K2 fused (round 4): the whole middle section of the frequency-domain gradient for one chain per block, in LDS --
// water level -> spectrum (k_rf_mid1) -> inverse real FFT -> rf(t), residual, misfit, weighted residual (k_rf_mid2) ->
// forward real FFT -> W_k for the adjoint sweep.  One launch and 4 KB of LDS per chain instead of five launches (two of
// them rocFFT's) that each wait for wave slots beside the surface-wave kernels: 1.35 -> ~0.4 ms of the RF stream inside a
// step (0.12 ms of kernel time either way).  rocFFT keeps librf's entries (B1), the forward-only calls and the time domain.
// The real transforms of length M = nft ride on a complex radix-2 FFT of length N = M / 2 (tw[j] = exp(-2 pi i j / M),
// j <= N: one table per nft, made on the host):
//   c2r  Z_k = (S_k + conj S_{N-k}) + i conj(tw_k) (S_k - conj S_{N-k}), z = N-point inverse FFT (unnormalised), x[2t] + i x[2t+1] = z_t
//   r2c  z_t = x[2t] + i x[2t+1], Z = N-point FFT, X_k = (Z_k + conj Z_{N-k}) / 2 - (i / 2) tw_k (Z_k - conj Z_{N-k}),  Z_N = Z_0
// -- the values rocFFT's c2r / r2c give (unnormalised, e^{-} forward) to rounding (2e-16 relative against numpy).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_fft_pow2(cplx* __restrict__ z, int N, int logN, const cplx* __restrict__ tw, int M, bool inverse)
{
    const int tid = threadIdx.x, nth = blockDim.x;
    for (int i = tid; i < N; i += nth) {                      // bit reversal
        const int r = (int)(__brev((unsigned)i) >> (32 - logN));
        if (i < r) { const cplx a = z[i]; z[i] = z[r]; z[r] = a; }
    }
    __syncthreads();
    for (int st = 1; st <= logN; st++) {
        const int len = 1 << st, half = len >> 1, step = M / len;
        for (int b = tid; b < (N >> 1); b += nth) {
            const int grp = b / half, j = b - grp * half;
            const int i0 = grp * len + j, i1 = i0 + half;
            cplx w = tw[j * step];                              // exp(-2 pi i j / len)
            if (inverse) w = conj(w);
            const cplx t = w * z[i1], u = z[i0];
            z[i0] = u + t; z[i1] = u - t;
        }
        __syncthreads();
    }
}

// gtab[k] = exp(-(w_k / 2 f0)^2) exp(-i w_k t0), etab[t] = exp(sigma (t dt - t0)) / dt: the chain-independent factors, made once
// per configuration (k_rf_mid_tables)
__global__ void k_rf_mid_tables(RfFreq f, cplx* __restrict__ gtab, double* __restrict__ etab)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < f.n2) {
        const double w = rf_wk(f, i);
        const double g = exp(-((w / 2 / f.f0) * (w / 2 / f.f0)));
        double s, c; sincos(w * f.t0, &s, &c);
        gtab[i] = C(g * c, -g * s);
    }
    if (i < f.nt) etab[i] = exp(f.sigma * (-f.t0 + i * f.dt)) / f.dt;
}

__global__ void __launch_bounds__(128)
k_rf_mid_fused(int n, RfFreq f, int logN, const RfLayer* __restrict__ lc, double* __restrict__ RR, double* __restrict__ wmax2,
               const cplx* __restrict__ tw, const cplx* __restrict__ gtab, const double* __restrict__ etab,
               const double* __restrict__ dobs, int ndata, double* __restrict__ dsyn,
               double* __restrict__ misfit_rf, cplx* __restrict__ Wout, const int* __restrict__ hi32,
               unsigned long long* __restrict__ stat32)
{
    extern __shared__ double lds_mid[];
    __shared__ double red[6][4];
    cplx* z = (cplx*)lds_mid;                                 // N + 1 complex numbers
    const int chain = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
    const int M = f.nft, N = M >> 1;
    double* rr = RR + (size_t)chain * 4 * f.n2p;
    double m1, m2;
    rf_mid_maxima(n, f, lc, rr, chain, hi32, stat32, red, m1, m2);
    if (tid == 0) wmax2[chain] = m2;
    // spectrum (RFModule.f90:393-401), Im of the DC and Nyquist bins dropped as FFTW's c2r does
    for (int k = tid; k <= N; k += nth) {
        cplx r21 = C(rr[k], rr[f.n2p + k]), r22 = C(rr[2 * f.n2p + k], rr[3 * f.n2p + k]);
        double wa = (r21 * conj(r21)).re;
        double fai = fmax(wa, f.water * m1);
        cplx S = (conj(r21) * r22) * (1.0 / fai) * gtab[k];
        if (k == 0 || k == N) S.im = 0.0;
        z[k] = S;
    }
    __syncthreads();
    // c2r: pairs (k, N - k) in place
    for (int k = tid; k <= (N >> 1); k += nth) {
        const int kb = N - k;
        const cplx a = z[k], b = z[kb];
        const cplx ac = conj(a), bc = conj(b);
        const cplx zk = (a + bc) + mul_i(conj(tw[k]) * (a - bc));
        const cplx zb = (b + ac) + mul_i(conj(tw[kb]) * (b - ac));
        z[k] = zk;
        if (kb != k && kb < N) z[kb] = zb;
    }
    __syncthreads();
    lds_fft_pow2(z, N, logN, tw, M, true);
    // rf(t) = x(t) / nft / dt * exp(sigma (t dt - t0)) (:404-407), residual, misfit, weighted residual -- packed for the r2c
    double acc = 0.0;
    for (int t2 = tid; t2 < N; t2 += nth) {
        const cplx v = z[t2];
        double wv[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int t = 2 * t2 + q;
            const double x = q ? v.im : v.re;
            wv[q] = 0.0;
            if (t < f.nt) {
                const double e = etab[t];                      // exp(sigma (t dt - t0)) / dt
                const double rf = x / f.nft * e;
                if (dsyn) dsyn[(size_t)chain * ndata + t] = rf;
                const double r = rf - dobs[t];
                acc += r * r;
                wv[q] = r * e;
            }
        }
        z[t2] = C(wv[0], wv[1]);
    }
    acc = wave_sum(acc);
    __syncthreads();
    if ((tid & 63) == 0) red[0][tid >> 6] = acc;
    __syncthreads();
    if (tid == 0 && misfit_rf) {
        double s = 0.0;
        for (int i = 0; i < (int)(nth >> 6); i++) s += red[0][i];
        misfit_rf[chain] = 0.5 * s;
    }
    lds_fft_pow2(z, N, logN, tw, M, false);
    // r2c post-processing -> W_k, k = 0 .. N
    cplx* W = Wout + (size_t)chain * f.n2;
    for (int k = tid; k <= N; k += nth) {
        const cplx a = z[k == N ? 0 : k], b = conj(z[(N - k) == N ? 0 : (N - k)]);
        const cplx x = (a + b) * 0.5 - mul_i(tw[k] * (a - b)) * 0.5;
        W[k] = x;
    }
}

// K2 mid 2: per chain -- rf(t) = irfft(S)(t)/dt * exp(sigma (t dt - t0)) (:404-407), residual,
// misfit, and the weighted residual whose forward FFT feeds the adjoint pass.
__global__ void __launch_bounds__(256)
k_rf_mid2(RfFreq f, const double* __restrict__ tser, const double* __restrict__ dobs, int ndata,
          double* __restrict__ dsyn, double* __restrict__ misfit_rf, double* __restrict__ wres)
{
    __shared__ double red[4];
    int chain = blockIdx.x, tid = threadIdx.x;
    const double* ts = tser + (size_t)chain * f.nft;
    double acc = 0.0;
    for (int t = tid; t < f.nft; t += blockDim.x) {
        double wv = 0.0;
        if (t < f.nt) {
            double e = exp(f.sigma * (-f.t0 + t * f.dt));
            double rf = ts[t] / f.nft / f.dt * e;
            if (dsyn) dsyn[(size_t)chain * ndata + t] = rf;
            if (dobs) {
                double r = rf - dobs[t];
                acc += r * r;
                wv = r / f.dt * e;
            }
        }
        if (wres) wres[(size_t)chain * f.nft + t] = wv;
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0 && misfit_rf) {
        double s = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) s += red[i];
        misfit_rf[chain] = 0.5 * s;
    }
}

// ---------------------------------------------------------------------------------------
// K1 pass B (adjoint): lane = frequency / lane = chain as in pass A.  Each lane forms its
// adjoint weights (u, v) from R21, R22, the water-levelled |R21^2|^2 and W = rfft(weighted
// residual), sweeps the column y top-down and emits Re(r_j . dA_j/dm . y_j) for the four
// parameter classes; a wave butterfly sums over the 64 frequencies of the wave and lane
// (j mod 64) keeps layer j's sum, so nothing is written until the sweep ends.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ V4 rf_adjoint_seed(const RfFreq& f, int k, cplx r21, cplx r22, cplx W, double wm2)
{
    double w = rf_wk(f, k);
    cplx sq = r21 * r21;
    double fai2 = fmax((sq * conj(sq)).re, f.water * wm2);
    double g = exp(-((w / 2 / f.f0) * (w / 2 / f.f0)));
    double s, c; sincos(w * f.t0, &s, &c);
    double ck = (k == 0 || k == f.n2 - 1) ? 1.0 : 2.0;
    cplx Q = (conj(sq) * C(c, -s)) * conj(W) * (ck * g / fai2 / f.nft);
    V4 y;
    if (f.rf_type == 1) { y.v[0] = -(Q * r22); y.v[1] = mul_i(Q * r21); }
    else { y.v[0] = -mul_i(Q * r21); y.v[1] = -(Q * r22); }
    y.v[2] = C(0.0); y.v[3] = C(0.0);
    return y;
}

// INV: the row of layer j is the row of layer j-1 times A_j^-1, starting from pass A's final row (RT: [chain][8][nkp]) --
// for the chains whose growth exponent allows it; the others read their stored rows (Rs).  With RT given both
// instantiations are launched and each takes its own chains (a block = one chain: the other kind leaves at once).
template <bool TAIL, bool INV = false, bool HST = false>      // HST: pass A stored exp / cos / sin of every (layer, frequency) (INV, not TAIL)
__global__ void __launch_bounds__(256)      // 238 VGPRs -> 2 waves/SIMD; forcing 3 or 4 (spills) measured no faster
k_rf_passB(int nchain, int n, RfFreq f, const RfLayer* __restrict__ lc, const double* __restrict__ RR,
           const double* __restrict__ Rs, const double* __restrict__ RT, const cplx* __restrict__ W,
           const double* __restrict__ wmax2, int npart, double* __restrict__ PG, unsigned* __restrict__ peel_resid,
           const int* __restrict__ slist, const int* __restrict__ scount, int* __restrict__ est_out,
           const double* __restrict__ Hs = nullptr)
{
  // slist (stored-row launch beside a peeling one): the blocks' y index strides over the chains pass A listed
  const int nsel = (!TAIL && slist) ? *scount : 1;
  if (!TAIL && est_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *est_out = nsel;   // (host-mapped: sizes the next launch)
  for (int sel = (!TAIL && slist) ? (int)blockIdx.y : 0; sel < nsel; sel += (!TAIL && slist) ? (int)gridDim.y : 1) {
    int chain, k, part;
    bool live = true;
    if (TAIL) {
        chain = blockIdx.x * blockDim.x + threadIdx.x; k = f.n2 - 1; part = npart - 1;
        if (chain >= nchain) return;
    } else {
        chain = slist ? slist[sel] : (int)blockIdx.y; k = blockIdx.x * blockDim.x + threadIdx.x;
        part = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        const int kmax = f.nk < f.n2 - 1 ? f.nk : f.n2 - 1;       // (the Nyquist bin has its own launch, TAIL)
        if (k >= kmax) { live = false; k = kmax - 1; }
    }
    const RfLayer* L = lc + (size_t)chain * n;
    const size_t n2p = f.n2p, nkp = f.nkp;
    const double* rr = RR + (size_t)chain * 4 * n2p + k;
    cplx r21 = C(rr[0], rr[n2p]), r22 = C(rr[2 * n2p], rr[3 * n2p]);
    cplx omega = C(rf_wk(f, k), -f.sigma), kk = f.p * omega;
    V4 y = rf_adjoint_seed(f, k, r21, r22, W[(size_t)chain * f.n2 + k], wmax2[chain]);
    if (!live) { y.v[0] = C(0.0); y.v[1] = C(0.0); }
    if (INV != (RT && (TAIL ? rf_growth_exponent(L, n, f.sigma, rf_wk(f, f.nk))
                            : rf_growth_exponent_wave(L, n, f.sigma, rf_wk(f, f.nk))) <= f.peel_emax)) continue;   // the other launch's chain
    const double* rs = (INV ? RT + (size_t)chain * 8 * nkp : Rs + ((size_t)chain * (n - 1)) * 8 * nkp) + k;
    double acc[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    const int lane = threadIdx.x & 63;
    double* pg = PG + ((size_t)chain * npart + part) * 4 * n;
    V4 r;
    if (INV) {
#pragma unroll
        for (int i = 0; i < 4; i++) r.v[i] = C(rs[(2 * i) * nkp], rs[(2 * i + 1) * nkp]);
    }
    const double* hs = HST ? Hs + ((size_t)chain * (n - 1)) * 6 * nkp + k : nullptr;
    for (int j = 0; j < n; j++) {
        cplx T[4];
        if (j < n - 1) {
            RfHyp H; RfA A;
            if (HST) {                             // pass A left the exponentials, cosines and sines of this (layer, frequency)
                const double* q = hs + (size_t)j * 6 * nkp;
                const RfHypB B{q[0], q[nkp], q[2 * nkp], q[3 * nkp], q[4 * nkp], q[5 * nkp]};
                rf_hyp_from(L[j], omega, B, H);
            } else rf_hyp(L[j], omega, H);
            if (INV) {
                rf_build_A(L[j], H, A);
                const V4 ra = r;                                  // the row above this layer (= r A)
                r = rf_row_times_Ainv(r, A);
                rf_layer_partials<false>(L[j], H, kk, r, y, T);
                const V4 ya = rf_A_times_col(A, y);
                T[0] = C(rf_rho_partial(L[j], ra, y, r, ya));
                y = ya;
            } else {
                const double* o = rs + (size_t)j * 8 * nkp;
#pragma unroll
                for (int i = 0; i < 4; i++) r.v[i] = C(o[(2 * i) * nkp], o[(2 * i + 1) * nkp]);
                rf_layer_partials(L[j], H, kk, r, y, T);
                rf_build_A(L[j], H, A);
                y = rf_A_times_col(A, y);
            }
        } else {
            rf_half_partials(L[j], omega, f.rf_type, y, T);
            if (INV && peel_resid) {
                // closure of the peeling: with every layer taken off, the row must be the half-space's own (rf_einv_row).
                // The largest relative miss of the call is kept (statistic "rf_peel_residual"): ~1e-14 where the waves
                // propagate; a post-critical slowness would show here
                const V4 e = rf_einv_row(L[j], f.rf_type);
                double d = 0.0, m = 0.0;
#pragma unroll
                for (int i = 0; i < 4; i++) { d += norm2(r.v[i] - e.v[i]); m += norm2(e.v[i]); }
                float q = (float)sqrt(d / m);
                if (!(q == q)) q = __builtin_inff();
                if (live) atomicMax(peel_resid, __float_as_uint(q));
            }
        }
        double v4[4];
#pragma unroll
        for (int ip = 0; ip < 4; ip++) {
            double v = T[ip].re;
            if (v != v) v = 0.0;                                  // NaN scrub (:698-703)
            v4[ip] = v;
            if (TAIL) pg[(size_t)ip * n + j] = v;
        }
        if (!TAIL) {
            double t4[4];
            wave_sum4_uniform(v4, t4);
#pragma unroll
            for (int ip = 0; ip < 4; ip++)
                if (lane == (j & 63)) acc[ip][j >> 6] = t4[ip];
        }
    }
    if (!TAIL) {
#pragma unroll
        for (int s = 0; s < 2; s++) {
            int j = s * 64 + lane;
            if (j < n) {
#pragma unroll
                for (int ip = 0; ip < 4; ip++) pg[(size_t)ip * n + j] = acc[ip][s];
            }
        }
    }
  }
}
// B1 kernel_all: materialise the partial spectra S_{p,j,k} (RFModule.f90:416-419) with one column sweep
// (the two unit-seed columns for R21_m and R22_m combined up front).  specp: [chain][4][n][n2] complex.
template <bool TAIL>
__global__ void __launch_bounds__(256)
k_rf_partial_spectra(int nchain, int n, RfFreq f, const RfLayer* __restrict__ lc, const double* __restrict__ RR,
                     const double* __restrict__ Rs, const double* __restrict__ wmax2, cplx* __restrict__ specp)
{
    int chain, k;
    if (TAIL) {
        chain = blockIdx.x * blockDim.x + threadIdx.x; k = f.n2 - 1;
        if (chain >= nchain) return;
    } else {
        chain = blockIdx.y; k = blockIdx.x * blockDim.x + threadIdx.x;
        if (k >= f.n2 - 1) return;
    }
    const RfLayer* L = lc + (size_t)chain * n;
    const size_t n2p = f.n2p;
    const double* rr = RR + (size_t)chain * 4 * n2p + k;
    cplx r21 = C(rr[0], rr[n2p]), r22 = C(rr[2 * n2p], rr[3 * n2p]);
    double w = rf_wk(f, k);
    cplx omega = C(w, -f.sigma), kk = f.p * omega;
    cplx sq = r21 * r21;
    double fai2 = fmax((sq * conj(sq)).re, f.water * wmax2[chain]);
    double g = exp(-((w / 2 / f.f0) * (w / 2 / f.f0)));
    double s, c; sincos(w * f.t0, &s, &c);
    cplx Q = (conj(sq) * C(c, -s)) * (g / fai2);
    const size_t nkp = f.nkp;                                   // (= n2p: this entry runs without the band limit)
    const double* rs = Rs + ((size_t)chain * (n - 1)) * 8 * nkp + k;
    const int c21 = (f.rf_type == 1) ? 0 : 1, c22 = 1 - c21;
    cplx* out = specp + (size_t)chain * 4 * n * f.n2 + k;
    // num = R22_m R21 - R21_m R22 is linear in the two unit-seed columns, and both are carried through the stack by
    // the same matrices: one sweep of the combined column  (+-i R21) e_c22 - R22 e_c21  yields num directly
    V4 y; y.v[0] = y.v[1] = y.v[2] = y.v[3] = C(0.0);
    y.v[c22] = (f.rf_type == 1) ? mul_i(r21) : -mul_i(r21);
    y.v[c21] = -r22;
    const bool edge = (k == 0 || k == f.n2 - 1);
    for (int j = 0; j < n; j++) {
        cplx T[4];
        if (j < n - 1) {
            const double* o = rs + (size_t)j * 8 * nkp;
            V4 r;
#pragma unroll
            for (int i = 0; i < 4; i++) r.v[i] = C(o[(2 * i) * nkp], o[(2 * i + 1) * nkp]);
            RfHyp H; RfA A;
            rf_hyp(L[j], omega, H);
            rf_layer_partials(L[j], H, kk, r, y, T);
            rf_build_A(L[j], H, A);
            y = rf_A_times_col(A, y);
        } else {
            rf_half_partials(L[j], omega, f.rf_type, y, T);
        }
#pragma unroll
        for (int ip = 0; ip < 4; ip++) {
            cplx t = T[ip];
            if (t.re != t.re || t.im != t.im) t = C(0.0);             // NaN scrub (:698-703)
            cplx S = Q * t;
            if (edge) S.im = 0.0;
            out[((size_t)ip * n + j) * f.n2] = S;
        }
    }
}

// scale batched inverse FFT output into kl[chain][4][n][nt]
__global__ void k_rf_scale_kl(size_t ntrace, RfFreq f, const double* __restrict__ tser, double* __restrict__ kl)
{
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ntrace * f.nt) return;
    size_t tr = g / f.nt; int t = (int)(g - tr * f.nt);
    kl[g] = tser[tr * f.nft + t] / f.nft / f.dt * exp(f.sigma * (-f.t0 + t * f.dt));
}

// ---------------------------------------------------------------------------------------
// K3 root search: lane = (sequence, chain); the whole per-period hand-over of the reference
// runs sequentially inside the lane, all 64 lanes evaluate the secular function together.
// Sequences: 0 = tRc, 1 = tRg, 2 = 1.05 tRg, 3 = 0.95 tRg (surfdisp.cpp:235-241).
// ---------------------------------------------------------------------------------------
struct SwdSeq { const double* t; int nper; double scale; int croot_off; int alt_vp; };   // croot_off in periods
struct SwdSeqs { SwdSeq s[4]; int nseq; int nper_total; };   // one wave family (Rayleigh or Love) per launch

// MODES: libsurf's `mode` argument > 0 -- the mode loop and its retry inside the state machine (RootSearchT<.., true>);
// craw = scratch of the unrounded roots c(k) of the mode before, laid out like croot; nmode = mode + 1.
struct SwdRootOut {             // outputs of one lane's search, readable (the retry pass looks for zeroed periods)
    double* cr; size_t stride; bool live;
    __device__ __forceinline__ void operator()(int k, double v) const { if (live) cr[(size_t)k * stride] = v; }
    __device__ __forceinline__ double get(int k) const { return live ? cr[(size_t)k * stride] : 1.0; }
};

template <bool LOVE, bool MODES>
__global__ void __launch_bounds__(64)
k_swd_roots(int nchain, int n, SwdSeqs Q, const float* __restrict__ mdl, double* __restrict__ croot,
            int* __restrict__ sflag, const int* __restrict__ list, const int* __restrict__ count,
            double* __restrict__ craw, int nmode)
{
    // list != nullptr: only the *count chains named there (the chains the warm start handed back, k_swd_warm)
    const int nsel = list ? *count : nchain;
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; (g & ~63) < Q.nseq * nsel; g += gridDim.x * blockDim.x) {
    int seq = g / (nsel > 0 ? nsel : 1), chain = g - seq * nsel;
    bool live = seq < Q.nseq;
    if (live && list) chain = list[chain];
    if (!live) { seq = 0; chain = 0; }
    const size_t s = (size_t)n * nchain;
    const SwdSeq sq = Q.s[seq];
    // Love group forward (_LoveGroup) searches with vp = 1.732 vs: array 4 of the Love model
    SwdModel M{mdl + chain, mdl + (LOVE && sq.alt_vp ? 4 : 1) * s + chain, mdl + 2 * s + chain, mdl + 3 * s + chain, nchain, n};
    const double* tp = sq.t; const double sc = sq.scale;
    auto T = [&](int k) { return tp[k] * sc; };
    const SwdRootOut out{croot + (size_t)sq.croot_off * nchain + chain, (size_t)nchain, live};
    RootSearchT<NevTabReg, MODES> rs;
    if (MODES) rs.set_modes(nmode, craw + (size_t)sq.croot_off * nchain + chain, (long)nchain);
    rs.begin(M, T, sq.nper);
    if (!live) rs.done = 1;
    while (__any(!rs.done)) {
        if (!rs.done) {
            double wvno = rs.omega / rs.creq;
            double del = LOVE ? swd_secular_love(M, wvno, rs.omega) : swd_secular(M, wvno, rs.omega);
            rs.advance(del, T, out);
        }
    }
    if (live) sflag[(size_t)seq * nchain + chain] = rs.flag;
    }
}

// K3 (split): G lanes per (sequence, chain).  Every lane of a group runs the same search state
// machine (identical inputs -> identical state, no broadcast needed); per secular-function
// evaluation the group's lanes share the layer loop: lane g builds the vector-independent part
// (15 numbers) of layers g, g+G, ... into LDS, then every lane runs the short sequential
// vector recurrence reading the group's entries back (LDS broadcast).  This cuts the serial
// instruction count per evaluation ~G-fold where it matters (sqrt/sincos/exp live in the
// layer part) and turns 128 latency-bound waves into 128*G waves.
// SPEC > 1 (small batches again): the block has SPEC wavefronts that all carry the same items and the same states.
// While an item steps through its scan (getsol's "move c by dc until the sign changes", surfdisp96.f:457-479) wavefront
// w evaluates the point the search WOULD ask for w steps ahead (RootSearch::scan_peek), so one round settles up to SPEC
// scan steps.  The results are then fed to the unchanged state machine in order, each only if the machine's next request
// is bit for bit the point that was evaluated; anything else (sign change, clamp, limit) discards the rest.  The
// sequence of (request, Delta) pairs the machine consumes is exactly that of the one-at-a-time search.
// F: the secular function (SwdRayFamily / SwdLoveFamily, swd_math.hpp)
template <class F, int LPL, int NSEG, int SPEC>   // LPL layers per lane in registers: (n-1) <= G*LPL;  NSEG segments;  SPEC wavefronts
__global__ void __launch_bounds__(64 * SPEC)
k_swd_roots_split(int nchain, int n, int G, SwdSeqs Q, const float* __restrict__ mdl,
                  const double* __restrict__ mdlc, double* __restrict__ croot, int* __restrict__ sflag,
                  const int* __restrict__ list, const int* __restrict__ count)
{
    extern __shared__ double split_lds[];        // per wavefront: entries [grp][m][15], NSEG > 1: rows [grp][chain][5];
    const int NG = 64 / G;                       // then SPEC > 1: Delta [SPEC][NG], the points they belong to [SPEC][NG]
    constexpr int NENT = F::NENT, NV = F::NV;
    constexpr int NCHAINS = 1 + NV * (NSEG - 1);
    const int sw = SPEC > 1 ? (int)(threadIdx.x >> 6) : 0;
    const size_t per_wave = (size_t)(n - 1) * NENT * NG + (size_t)(NCHAINS + 1) * NV * NG;   // (+1: the half-space vector)
    double* ent_lds = split_lds + (size_t)sw * per_wave;
    double* seg_lds = ent_lds + (size_t)(n - 1) * NENT * NG;
    double* del_lds = split_lds + (size_t)SPEC * per_wave;
    double* pt_lds = del_lds + SPEC * NG;
    const int seglen = (n - 1 + NSEG - 1) / NSEG;      // layers per segment (the shallowest one may be shorter)
    const int lane = threadIdx.x & 63, grp = lane / G, lg = lane - grp * G;
    // a group's numbers are contiguous, so that every read below is one base register + an immediate offset
    double* const ent_g = ent_lds + (size_t)grp * (n - 1) * NENT;
    double* const seg_g = seg_lds + (size_t)grp * (NCHAINS + 1) * NV;
    double* const hs_g = seg_g + NCHAINS * NV;   // half-space start vector, built beside the layer entries by the group's last lane
    // list != nullptr: only the *count chains named there (the chains the warm start handed back, k_swd_warm); the
    // blocks then stride over the items, whose number the host does not know
    const int nsel = list ? *count : nchain;
    for (int blk = blockIdx.x; blk * NG < Q.nseq * nsel; blk += gridDim.x) {
    int item = blk * NG + grp;                   // (sequence, chain) handled by this group
    int seq = item / (nsel > 0 ? nsel : 1), chain = item - seq * nsel;
    bool live = seq < Q.nseq;
    if (live && list) chain = list[chain];
    if (!live) { seq = 0; chain = 0; }
    const size_t s = (size_t)n * nchain;
    const SwdSeq sq = Q.s[seq];
    // Love group forward (_LoveGroup) searches with vp = 1.732 vs: array 4 of the Love model
    SwdModel M{mdl + chain, mdl + (F::LOVE && sq.alt_vp ? 4 : 1) * s + chain, mdl + 2 * s + chain, mdl + 3 * s + chain, nchain, n};
    const double* tp = sq.t; const double sc = sq.scale;
    auto T = [&](int k) { return tp[k] * sc; };
    double* cr = croot + (size_t)sq.croot_off * nchain + chain;
    const bool writer = live && lg == 0;
    auto out = [&](int k, double v) { if (writer) cr[(size_t)k * nchain] = v; };
    RootSearch rs;
    rs.begin(M, T, sq.nper);
    if (!live) rs.done = 1;
    const double* lc0 = mdlc + chain;
    auto loadL = [&](int m) {
        const double* o = lc0 + (size_t)m * 6 * nchain;
        return SwdLayerC{o[0], o[(size_t)nchain], o[(size_t)2 * nchain], o[(size_t)3 * nchain],
                         o[(size_t)4 * nchain], o[(size_t)5 * nchain]};
    };
    const SwdLayerC Lhalf = loadL(n - 1);
    SwdLayerC Lmine[LPL];                        // this lane's layers lg, lg+G, ... stay in registers
#pragma unroll
    for (int q = 0; q < LPL; q++) { int m = lg + q * G; Lmine[q] = loadL(m < n - 1 ? m : n - 2); }
    while (__any(!rs.done)) {                    // (the wavefronts of a block hold identical states: same trip count)
        double omega = rs.omega < 1.0e-4 ? 1.0e-4 : rs.omega;
        double creq = rs.creq;
        bool act = !rs.done;
        if (SPEC > 1 && sw > 0) act = act && rs.scan_peek(sw, creq);
        double wvno = rs.omega / creq, wvno2 = wvno * wvno, iomega = 1.0 / omega;
        double delta = 0.0;
        if (act) {
#pragma unroll
            for (int q = 0; q < LPL; q++) {
                int m = lg + q * G;
                if (m < n - 1) {
                    double ent[NENT];
                    F::entries(Lmine[q], wvno, wvno2, omega, iomega, ent);
#pragma unroll
                    for (int i = 0; i < NENT; i++) ent_g[m * NENT + i] = ent[i];
                }
            }
            if (lg == G - 1) {               // the lane with the fewest layers (none when G > n - 1)
                double e[NV];
                F::halfspace(Lhalf, wvno, wvno2, omega, iomega, e);
#pragma unroll
                for (int j = 0; j < NV; j++) hs_g[j] = e[j];
            }
        }
        __syncthreads();
        const double tt = -2.0 * wvno2;
        if constexpr (NSEG == 1) {
            if (act) {
                double e[NV];
#pragma unroll
                for (int j = 0; j < NV; j++) e[j] = hs_g[j];
                const double* pe = ent_g + (size_t)(n - 2) * NENT;
                for (int m = n - 2; m >= 0; m--, pe -= NENT) {
                    double cur[NENT];
#pragma unroll
                    for (int i = 0; i < NENT; i++) cur[i] = pe[i];
                    F::apply(e, cur, tt);
                    if ((m & 7) == 0) swd_rescale_pow2_n<NV>(e);
                }
                delta = swd_finish_n<NV>(e);
            }
        } else {
            // lane lg < NCHAINS runs chain lg: chain 0 = half-space vector through the deepest segment, chain 1 + NV (s - 1) + i
            // = unit vector i through segment s (s = 1 .. NSEG - 1, counted from the deepest)
            if (act && lg < NCHAINS) {
                const int sg = lg == 0 ? 0 : 1 + (lg - 1) / NV, ui = lg == 0 ? -1 : (lg - 1) % NV;
                double e[NV];
#pragma unroll
                for (int j = 0; j < NV; j++) e[j] = lg == 0 ? hs_g[j] : ((j == ui) ? 1.0 : 0.0);
                int mhi = n - 2 - sg * seglen, mlo = mhi - seglen + 1;
                if (mlo < 0) mlo = 0;
                // two layers per trip, the next layer's entries on their way while this one is applied (the LDS round trip
                // is as long as the 25 FMAs); reading one layer past the segment's end is harmless (clamped to layer 0)
                const double* pe = ent_g + (size_t)mhi * NENT;
                double ca[NENT], cb[NENT];
#pragma unroll
                for (int i = 0; i < NENT; i++) ca[i] = pe[i];
                for (int m = mhi; m >= mlo; m -= 2) {
                    const double* pb = pe - (m - 1 >= 0 ? NENT : 0);
#pragma unroll
                    for (int i = 0; i < NENT; i++) cb[i] = pb[i];
                    F::apply(e, ca, tt);
                    pe = pb - (m - 2 >= 0 ? NENT : 0);
#pragma unroll
                    for (int i = 0; i < NENT; i++) ca[i] = pe[i];
                    if (m - 1 >= mlo) F::apply(e, cb, tt);
                }
#pragma unroll
                for (int j = 0; j < NV; j++) seg_g[lg * NV + j] = e[j];
            }
            __syncthreads();
            if (act) {
                double e[NV];
#pragma unroll
                for (int j = 0; j < NV; j++) e[j] = seg_g[j];
                swd_rescale_pow2_n<NV>(e);
#pragma unroll
                for (int sg = 1; sg < NSEG; sg++) {
                    if (n - 2 - sg * seglen < 0) break;                  // fewer layers than segments
                    double nw[NV];
#pragma unroll
                    for (int j = 0; j < NV; j++) nw[j] = 0.0;
#pragma unroll
                    for (int i = 0; i < NV; i++) {
#pragma unroll
                        for (int j = 0; j < NV; j++)
                            nw[j] += e[i] * seg_g[(1 + NV * (sg - 1) + i) * NV + j];
                    }
#pragma unroll
                    for (int j = 0; j < NV; j++) e[j] = nw[j];
                    if (sg & 1) swd_rescale_pow2_n<NV>(e);   // the range is safe for two segments at a time
                }
                delta = swd_finish_n<NV>(e);
            }
        }
        if constexpr (SPEC == 1) {
            if (act) rs.advance(delta, T, out);
        } else {
            if (lg == 0) { del_lds[sw * NG + grp] = delta; pt_lds[sw * NG + grp] = act ? creq : __longlong_as_double(0x7ff8000000000000LL); }
            __syncthreads();
            if (!rs.done) {
                rs.advance(del_lds[grp], T, out);
                bool on = true;
#pragma unroll
                for (int w = 1; w < SPEC; w++) {
                    // the machine now asks for rs.creq: use wavefront w's result only if that is the very point it evaluated
                    on = on && !rs.done && rs.phase == RootSearch::PH_SCAN && pt_lds[w * NG + grp] == rs.creq;
                    if (on) rs.advance(del_lds[w * NG + grp], T, out);
                }
            }
        }
        __syncthreads();
    }
    if (writer) sflag[(size_t)seq * nchain + chain] = rs.flag;
    }
}

// ---------------------------------------------------------------------------------------
// K3w warm-started root refinement inside a trajectory: lane = (item, chain), item = (sequence, period) of one wave
// family -- every period on its own, no hand-over between periods (WarmSearch, swd_math.hpp).  The previous evaluation
// of the SAME chains left its roots (croot), kernels (krn) and model behind; k_prep_joint wrote the model change dxT.
// A lane predicts its root to first order, brackets it inside the trust radius, refines it by false position and
// overwrites croot in place.  A lane that cannot (no usable previous evaluation, no sign change where the first-order
// model says, root above the fastest layer) puts its CHAIN on the list for the reference-semantics search that
// follows on the same stream (k_swd_roots_split / k_swd_roots with list), which rewrites all of that chain's roots
// and alone decides its flag.
// ---------------------------------------------------------------------------------------
struct SwdWarm {
    const double* dxT;      // [2n][chain] model change since the previous evaluation (vs, thk)
    const int* valid;       // [chain] the previous evaluation of this chain succeeded
    const int* force;       // [chain] the caller wants the reference-semantics search this time (or nullptr)
    int* need;              // [chain] out: 1 = goes to the reference-semantics search
    int* count; int* list;  // ... and the compacted list of those chains
    unsigned long long* stats;   // [0] chains handed back, [1] secular evaluations, [2] items refined
    unsigned char* sgn;     // [item][chain] sign bit of the secular function just below the refined root (2: no root)
    int* irr; int* icount; int* ilist;   // chains with an irregular sequence (k_swd_warm_check -> k_swd_warm_walk)
    int* count2; int* list2;             // chains handed back by the branch test (the search of `list` is under way by then)
    int* wide;              // [chain] a first-order change above WARM_L1MAX somewhere: every sequence of the chain walks the grid
    double* slope;          // [item][chain] d(secular)/dc at the root as the last warm search of the item left it (0 = unknown)
    float* betmx;           // [2][chain] fastest S velocity of the Rayleigh / Love search model (k_swd_warm -> k_swd_warm_check)
    double* cwarm;          // [item][chain] the warm-started roots as k_swd_warm left them (k_swd_exact reads them while it overwrites croot), or nullptr
    int* count3; int* list3;             // chains handed back by k_swd_exact
    const int* pend;        // [chain] 1: the chain was handed back in the step before and its search ran in the background: croot holds its roots for THIS model
    unsigned char* sg1;     // [2][4][chain] sign bit of the sequence's first evaluation (del1st), left by the first-period walk for the dense walk of the later periods
    // flow entries: the chains' trajectory state as the step found it (nullptr elsewhere).  An IDLE chain -- waiting for the host
    // after a trajectory, or failed -- has not moved and nothing reads its evaluation (k_flow_post): it is neither continued nor
    // handed back (a failed chain has no valid previous evaluation and would go to the full search at every step it waits --
    // the slowest searches there are, for nothing; and in the background form nothing would order such a search against the
    // step that evaluates the chain's NEXT start model)
    const int* f_rem; const int* f_fresh; const int* f_ok;
    const double* walk_roots;   // [item][chain] what the grid walks take for the continued roots: cwarm beside k_swd_exact (which overwrites croot meanwhile), else croot
    int widen;              // option "swd_warm_widen": 1 = the search may go on beyond the trust radius (WarmSearch::wide)
    // Error feedback of the predictor (round 5, option "swd_warm_feedback"): what the first-order prediction missed by at the
    // previous step -- root - (previous root + G . dx), the second-order term of the root along the trajectory -- is added to
    // this step's prediction: inside a trajectory consecutive moves dt M^-1 p are nearly equal, and so are their second-order
    // terms.  [item][chain]; zero where there is nothing to carry over (a trajectory that starts with this step, a failed search).
    double* ferr;
    // "swd_walk_window" (round 6): < 0 = every period of a sequence with anomalous dispersion somewhere walks the reference's grid;
    // W >= 0 = only the periods within W of an anomalous one (the pair j - 1, j with c(j) <= c(j - 1) - 1.5 dc counts for both)
    // do, the others take the regular sequences' test -- one evaluation at the point their scan starts from
    int walk_window;
    // "swd_cold_first" (round 6, batches of a few chains): every chain takes the search without a prediction (k_swd_cold_scan) --
    // k_swd_warm only sorts out the idle chains and lists the others
    int decline_all;
    // ... and in the larger of those small batches, a chain that ended the evaluation BEFORE this one on the hand-back list (a wild
    // chain is wild for many steps: the branch test declines its continued roots again and again, 2 ms of sequential search each
    // time) goes straight to the search without a prediction this time.  [chain] the previous evaluation's `need`, or nullptr.
    const int* need_prev;
    // (what counts is the chain's last EVALUATION, not the device step before this one -- a chain that waited a step for the host has
    // a 2 there, and how often it waits is a matter of timing, which no result may depend on: [chain] the flag as the chain's last
    // evaluation left it, kept here across the steps it sits out; nullptr: off)
    int* cold_again;
};
// is period k of a sequence (roots cq, stride nchain) within `win` periods of an anomalous pair?
__device__ __forceinline__ bool swd_walk_near(const double* cq, size_t nchain, int nper, int k, int win, double dcs) {
    if (win < 0) return true;
    const int j0 = max(1, k - win), j1 = min(nper - 1, k + win + 1);
    bool near = false;
    for (int j = j0; j <= j1; j++) near = near || (cq[(size_t)(j - 1) * nchain] - 1.5 * dcs >= cq[(size_t)j * nchain]);
    return near;
}

// Searches a round of k_swd_warm did not finish within its budget of evaluations: [field][slot], slot = position in the round's
// list.  A wavefront executes what its SLOWEST lane needs, and the searches are very uneven -- 2 evaluations where the Newton start
// brackets the root at once (most items), 3-6 through a bracket, 20-60 through a widened bracket and bisection (a few per
// thousand): measured on the bench's chains, 2.56 evaluations per lane but 7.98 executed per lane.  So the search runs in ROUNDS:
// every lane gets a budget; what is unfinished then is written out densely (wavefront-aggregated slots) and the next round picks
// it up 64 unfinished searches to a wavefront.  A lane's sequence of evaluations does not depend on who shares its wavefront:
// the results are the single-round kernel's bit for bit.
struct WarmSpill {
    double* d;                    // [WARM_SPILL_ND][cap]
    unsigned long long* bits;     // [cap] the machine's small integers, packed
    unsigned long long* item;     // [cap] the lane's global item index (~0: not a slot)
    int* count;                   // slots handed out (may exceed cap: the lanes beyond finish in place)
    int cap;
};
constexpr int WARM_SPILL_ND = 17;

// What a finished warm search leaves behind (k_swd_warm, k_swd_warm_coop): the root rounded to float32 like the reference's, the
// sign below it for the branch test, its slope and the predictor's miss for the next step -- or the chain on the hand-back list.
// cause: 4 no usable previous evaluation / forced, 5 step too large for a first-order model, 6 no sign change inside the trust
// radius or no convergence, 7 root above the fastest layer  (statistics only)
__device__ __forceinline__ void swd_warm_decline(const SwdWarm& W, int chain, int cause) {
    if (atomicExch(&W.need[chain], 1) == 0) {
        W.list[atomicAdd(W.count, 1)] = chain;
        atomicAdd(&W.stats[0], 1ull);
        atomicAdd(&W.stats[cause], 1ull);
    }
}
__device__ __forceinline__ bool swd_warm_finish(const WarmSearch& ws, bool fin, bool refused, float betmx, int e, int chain, int nchain,
                                                double cprev, double dc, double* __restrict__ croot, const SwdWarm& W) {
    const bool ok = fin && ws.phase == WarmSearch::W_DONE && !(ws.root > (double)betmx);      // getsol :483-485
    // a root found beyond the trust radius: the grid walk has the word (statistic 13 counts the chains, as for large moves)
    if (ok && ws.wide() && atomicExch(&W.wide[chain], 1) == 0) atomicAdd(&W.stats[13], 1ull);
    if (!fin) {}
    else if (ok) {
        croot[(size_t)e * nchain + chain] = (double)(float)ws.root;                    // surfdisp96.f:302
        if (W.cwarm) W.cwarm[(size_t)e * nchain + chain] = (double)(float)ws.root;
        W.sgn[(size_t)e * nchain + chain] = signbit(ws.fa) ? 1 : 0;                    // (a, fa): the bracket's lower end
    } else {
        swd_warm_decline(W, chain, refused ? 5 : (ws.phase == WarmSearch::W_DONE ? 7 : 6));
        // (diagnostics: why a search failed -- 24: no sign change out to the widest bracket, 25: anything else)
        if (!refused && ws.phase != WarmSearch::W_DONE) atomicAdd(&W.stats[ws.eps >= fmin(WARM_RWIDE * ws.R, fmax(ws.R, WARM_RWIDE_ABS)) ? 24 : 25], 1ull);
    }
    if (fin) W.slope[(size_t)e * nchain + chain] = ok ? ws.slope : 0.0;
    if (fin && W.ferr) {
        // a trajectory that starts with this step has not moved (and its next move has a new momentum): nothing to carry over
        const bool fresh = W.f_fresh && W.f_fresh[chain];
        W.ferr[(size_t)e * nchain + chain] = (ok && !fresh && dc == dc) ? ws.root - (cprev + dc) : 0.0;
    }
    return ok;
}
// a search's state out of / into a WarmSpill slot
__device__ __forceinline__ void swd_warm_load(const WarmSpill& in, size_t sl, WarmSearch& ws, double& cprev, double& dc, double& l1, double& fb,
                                              float& betmx, int& attempt, int& nev_first) {
    const size_t cp = (size_t)in.cap;
    const double* D = in.d + sl;
    ws.cpred = D[0]; ws.eps = D[cp]; ws.R = D[2 * cp]; ws.a = D[3 * cp]; ws.fa = D[4 * cp]; ws.b = D[5 * cp]; ws.fb = D[6 * cp];
    ws.creq = D[7 * cp]; ws.root = D[8 * cp]; ws.slope = D[9 * cp]; ws.f0 = D[10 * cp]; ws.mlast = D[11 * cp];
    cprev = D[12 * cp]; dc = D[13 * cp]; l1 = D[14 * cp]; fb = D[15 * cp]; betmx = (float)D[16 * cp];
    ws.unpack_small(in.bits[sl], attempt, nev_first);
}

template <class F, bool SPH, bool FIRST>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3)))          // (round 5's bisection / wide-bracket paths took it to 171 VGPRs: three wavefronts per SIMD end at 168)
k_swd_warm(int nchain, int n, SwdSeqs Q, const double* __restrict__ mdlc, const double* __restrict__ sph,
           const double* __restrict__ krn, const double* __restrict__ ugr, size_t ntot, double* __restrict__ croot, SwdWarm W,
           WarmSpill in, WarmSpill out, int budget, int round)
{
  const int nin = FIRST ? 0 : min(*in.count, in.cap);
  for (size_t base = (size_t)blockIdx.x * 64; FIRST ? base == (size_t)blockIdx.x * 64 : base < (size_t)nin; base += (size_t)gridDim.x * 64) {
    const size_t slot = base + threadIdx.x;
    size_t g = slot;
    // (no lane leaves before the wavefront's statistics at the end: `live` instead of early returns)
    bool live = FIRST ? g < (size_t)Q.nper_total * nchain : slot < (size_t)nin;
    if (!FIRST) { g = live ? (size_t)in.item[slot] : 0; live = live && g != ~0ull; if (!live) g = 0; }
    const int el = live ? (int)(g / nchain) : 0, chain = live ? (int)(g - (size_t)el * nchain) : 0;
    const int e = Q.s[0].croot_off + el;
    int seq = 0;
    while (seq + 1 < Q.nseq && e >= Q.s[seq + 1].croot_off) seq++;
    const int k = e - Q.s[seq].croot_off;
    const size_t s = (size_t)n * nchain;
    auto decline = [&](int cause) { swd_warm_decline(W, chain, cause); };
    WarmSearch ws;
    double cprev = 0.0, dc = 0.0, l1 = 0.0, fb = 0.0;
    float betmx = -1.e20f;
    bool refused = false;
    int nev_first = 0, attempt = 0;
    if (FIRST) {
        // a chain whose search of the step before ran in the background: nothing to do here, and nothing for the branch test
        // or the reference-root stage either (they skip chains with a flag)
        if (live && W.pend && W.pend[chain]) { W.need[chain] = 2; live = false; }
        if (live && W.f_rem && !W.f_fresh[chain] && (W.f_rem[chain] <= 0 || !W.f_ok[chain])) { W.need[chain] = 2; live = false; }    // idle
        if (live) W.sgn[(size_t)e * nchain + chain] = 2;
        bool again = false;
        if (W.cold_again && g < (size_t)Q.nper_total * nchain) {
            const int np_ = W.need_prev ? W.need_prev[chain] : 0;           // (no warm-started evaluation before this one: start afresh)
            again = np_ == 2 ? W.cold_again[chain] != 0 : np_ == 1;
            if (el == 0 && np_ != 2) W.cold_again[chain] = again ? 1 : 0;     // (readers of the old value are the lanes that saw a 2: no write then)
        }
        if (live && (!W.valid[chain] || (W.force && W.force[chain]) || W.decline_all || again)) { decline(4); live = false; }
        // first-order prediction from the previous model's kernels (model_surf.py:184's chain rule; thickness kernel =
        // suffix sum of the interface partials, sregn96.f90:1727-1731); SPH: kernels of the flattened model mapped with
        // vtp / dtp / rtp as swd_kernel_value does
        const double* kr0 = krn + (size_t)e * 4 * s + chain;
        const double ksc = ugr[ntot + (size_t)e * nchain + chain], kfac = ugr[2 * ntot + (size_t)e * nchain + chain];    // swd_krn's scales
        double suf = 0.0;
        for (int m = n - 1; m >= 0; m--) {
            const size_t lm = (size_t)m * nchain;
            // (chain-ruled storage: slot 0 = d c / d vs of the previous model, its flattening factors inside; slot 1 = the interface partial)
            const double gv = kr0[lm] * ksc;
            double kh = kr0[s + lm] * kfac;
            if (fabs(kh) < 1.0e-38) kh = 0.0;
            if (SPH) kh *= sph[5 * s + lm + chain];
            const double t1 = gv * W.dxT[lm + chain], t2 = suf * W.dxT[s + lm + chain];
            dc += t1 + t2; l1 += fabs(t1) + fabs(t2);
            suf += kh;
            betmx = fmaxf(betmx, (float)mdlc[((size_t)m * 6 + 3) * nchain + chain]);
        }
        cprev = croot[(size_t)e * nchain + chain];
        if (live && k == 0) W.betmx[(size_t)(F::LOVE ? 1 : 0) * nchain + chain] = betmx;
        // (a correction larger than the first-order terms themselves is not a second-order term: ignored)
        fb = (W.ferr && live) ? W.ferr[(size_t)e * nchain + chain] : 0.0;
        if (!(fabs(fb) <= l1)) fb = 0.0;
        ws.begin(cprev, dc + fb, l1, W.slope[(size_t)e * nchain + chain]);
        if (W.widen && cprev > 0.0 && (!(dc == dc) || !(l1 <= WARM_L1WIDE))) ws.begin_wide(cprev);
        else if (!(dc == dc) || !(l1 == l1)) ws.phase = WarmSearch::W_FAIL;
        if (!live) { ws.phase = WarmSearch::W_FAIL; ws.nev = 0; }
        if (live && !(l1 <= WARM_L1MAX) && ws.active() && atomicExch(&W.wide[chain], 1) == 0) atomicAdd(&W.stats[13], 1ull);
        refused = !ws.active();
    } else {
        swd_warm_load(in, live ? slot : 0, ws, cprev, dc, l1, fb, betmx, attempt, nev_first);
        if (!live) { ws.phase = WarmSearch::W_FAIL; ws.nev = 0; }
    }
    const double omega = (2.0 * 3.141592653589793) / (Q.s[seq].t[k] * Q.s[seq].scale);
    const double* lc0 = mdlc + chain;
    auto loadL = [&](int m) {
        const double* o = lc0 + (size_t)m * 6 * nchain;
        return SwdLayerC{o[0], o[(size_t)nchain], o[(size_t)2 * nchain], o[(size_t)3 * nchain],
                         o[(size_t)4 * nchain], o[(size_t)5 * nchain]};
    };
    int used = 0, left = budget;
    bool spilled = false;
    for (;;) {
        while (__any(ws.active() && left > 0)) {
            if (ws.active() && left > 0) {
                ws.advance(swd_secular_family<F>(n, loadL, omega, ws.creq), W.widen != 0);
                used++; left--;
                // a search that failed from a corrected prediction starts over from the plain first-order one (the correction is a
                // guess: where the trajectory has just turned, or next to another mode, it points the wrong way)
                if (attempt == 0 && live && !refused && fb != 0.0 && ws.phase == WarmSearch::W_FAIL) {
                    nev_first = ws.nev; fb = 0.0; attempt = 1; ws.begin(cprev, dc, l1, 0.0);
                }
            }
        }
        // out of budget: the unfinished searches of this wavefront go to the next round, densely
        const bool unfinished = ws.active();
        const unsigned long long um = __ballot(unfinished);
        if (um == 0ull) break;
        int sbase = 0;
        const int nun = __popcll(um);
        if ((threadIdx.x & 63) == 0) sbase = atomicAdd(out.count, nun);
        sbase = __shfl(sbase, 0, 64);
        if (sbase + nun > out.cap) {
            // no room: the slots handed out stay empty, the searches finish here
            if (unfinished) { const int sl = sbase + __popcll(um & ((1ull << (threadIdx.x & 63)) - 1ull)); if (sl < out.cap) out.item[sl] = ~0ull; }
            left = 0x7fffffff;
            continue;
        }
        if (unfinished) {
            const size_t sl = (size_t)sbase + __popcll(um & ((1ull << (threadIdx.x & 63)) - 1ull)), cp = (size_t)out.cap;
            double* D = out.d + sl;
            D[0] = ws.cpred; D[cp] = ws.eps; D[2 * cp] = ws.R; D[3 * cp] = ws.a; D[4 * cp] = ws.fa; D[5 * cp] = ws.b; D[6 * cp] = ws.fb;
            D[7 * cp] = ws.creq; D[8 * cp] = ws.root; D[9 * cp] = ws.slope; D[10 * cp] = ws.f0; D[11 * cp] = ws.mlast;
            D[12 * cp] = cprev; D[13 * cp] = dc; D[14 * cp] = l1; D[15 * cp] = fb; D[16 * cp] = (double)betmx;
            out.bits[sl] = ws.pack_small(attempt, nev_first);
            out.item[sl] = (unsigned long long)g;
            spilled = true;
        }
        break;
    }
    const bool fin = live && !spilled;
    ws.nev += nev_first;
    const bool ok = swd_warm_finish(ws, fin, refused, betmx, e, chain, nchain, cprev, dc, croot, W);
    int nev = fin ? ws.nev : 0, nok = ok ? 1 : 0, nmax = used, nlive = fin ? 1 : 0, nsp = spilled ? 1 : 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        nev += __shfl_xor(nev, off, 64); nok += __shfl_xor(nok, off, 64);
        nmax = max(nmax, __shfl_xor(nmax, off, 64)); nlive += __shfl_xor(nlive, off, 64); nsp += __shfl_xor(nsp, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&W.stats[1], (unsigned long long)nev); atomicAdd(&W.stats[2], (unsigned long long)nok);
        // divergence of the search: evaluations of the lanes that finished here (26), of this wavefront's slowest lane in this
        // round (27: what the wavefront executes is 64 x this), lanes that searched (28), searches passed on by round 1 / 2 / 3 (29-31)
        atomicAdd(&W.stats[26], (unsigned long long)nev); atomicAdd(&W.stats[27], (unsigned long long)nmax);
        atomicAdd(&W.stats[28], (unsigned long long)nlive);
        if (nsp && round >= 0 && round < 3) atomicAdd(&W.stats[29 + round], (unsigned long long)nsp);
    }
  }
}

// The LAST round of the warm search (what two budgets have not finished: a few per cent of the items, with 5 to 60 evaluations
// still to go -- and the stage ends with its slowest search): G = 16 lanes per search.  Every lane of a group runs the same
// machine on the same numbers (identical inputs -> identical states, as in k_swd_roots_split); per evaluation lane j builds the
// vector-independent numbers of layers j, j + 16, ... into LDS, then every lane runs the short vector recurrence over them.
// The arithmetic is swd_secular_family's operation for operation (that function IS this evaluation done by one lane), so the
// roots are the single-lane search's bit for bit -- in a third of the time per evaluation.  n - 1 <= 64 layers.
template <class F>
__global__ void __launch_bounds__(64)
k_swd_warm_coop(int nchain, int n, SwdSeqs Q, const double* __restrict__ mdlc, double* __restrict__ croot, SwdWarm W, WarmSpill in)
{
    constexpr int G = 16, NG = 64 / G, LPL = 4, NENT = F::NENT, NV = F::NV;
    extern __shared__ double coop_lds[];         // per group: entries [m][NENT], then the half-space vector [NV]
    const int lane = threadIdx.x & 63, grp = lane / G, lg = lane - grp * G;
    double* const ent_g = coop_lds + (size_t)grp * ((size_t)(n - 1) * NENT + NV);
    double* const hs_g = ent_g + (size_t)(n - 1) * NENT;
    const int nin = min(*in.count, in.cap);
    for (int blk = blockIdx.x; blk * NG < nin; blk += gridDim.x) {
        const int slot = blk * NG + grp;
        bool live = slot < nin;
        size_t g = live ? (size_t)in.item[slot] : 0;
        live = live && g != ~0ull;
        if (!live) g = 0;
        const int el = (int)(g / nchain), chain = (int)(g - (size_t)el * nchain);
        const int e = Q.s[0].croot_off + el;
        int seq = 0;
        while (seq + 1 < Q.nseq && e >= Q.s[seq + 1].croot_off) seq++;
        const int k = e - Q.s[seq].croot_off;
        WarmSearch ws;
        double cprev = 0.0, dc = 0.0, l1 = 0.0, fb = 0.0;
        float betmx = -1.e20f;
        int nev_first = 0, attempt = 0;
        swd_warm_load(in, live ? (size_t)slot : 0, ws, cprev, dc, l1, fb, betmx, attempt, nev_first);
        if (!live) { ws.phase = WarmSearch::W_FAIL; ws.nev = 0; }
        const double omega_raw = (2.0 * 3.141592653589793) / (Q.s[seq].t[k] * Q.s[seq].scale);
        const double omega = omega_raw < 1.0e-4 ? 1.0e-4 : omega_raw, iomega = 1.0 / omega;
        const double* lc0 = mdlc + chain;
        auto loadL = [&](int m) {
            const double* o = lc0 + (size_t)m * 6 * nchain;
            return SwdLayerC{o[0], o[(size_t)nchain], o[(size_t)2 * nchain], o[(size_t)3 * nchain],
                             o[(size_t)4 * nchain], o[(size_t)5 * nchain]};
        };
        const SwdLayerC Lhalf = loadL(n - 1);
        SwdLayerC Lmine[LPL];                    // this lane's layers lg, lg + G, ... stay in registers
#pragma unroll
        for (int q = 0; q < LPL; q++) { const int m = lg + q * G; Lmine[q] = loadL(m < n - 1 ? m : n - 2); }
        int used = 0;
        while (__any(ws.active())) {
            const bool act = ws.active();
            const double wvno = omega_raw / ws.creq, wvno2 = wvno * wvno, tt = -2.0 * wvno2;
            if (act) {
#pragma unroll
                for (int q = 0; q < LPL; q++) {
                    const int m = lg + q * G;
                    if (m < n - 1) {
                        double ent[NENT];
                        F::entries(Lmine[q], wvno, wvno2, omega, iomega, ent);
#pragma unroll
                        for (int i = 0; i < NENT; i++) ent_g[m * NENT + i] = ent[i];
                    }
                }
                if (lg == G - 1) {
                    double eh[NV];
                    F::halfspace(Lhalf, wvno, wvno2, omega, iomega, eh);
#pragma unroll
                    for (int j = 0; j < NV; j++) hs_g[j] = eh[j];
                }
            }
            __syncthreads();
            double delta = 0.0;
            if (act) {
                double ev[NV];
#pragma unroll
                for (int j = 0; j < NV; j++) ev[j] = hs_g[j];
                const double* pe = ent_g + (size_t)(n - 2) * NENT;
                for (int m = n - 2; m >= 0; m--, pe -= NENT) {
                    double cur[NENT];
#pragma unroll
                    for (int i = 0; i < NENT; i++) cur[i] = pe[i];
                    F::apply(ev, cur, tt);
                    if ((m & 7) == 0) swd_rescale_pow2_n<NV>(ev);
                }
                delta = swd_finish_n<NV>(ev);
            }
            __syncthreads();
            if (act) {
                ws.advance(delta, W.widen != 0);
                used++;
                if (attempt == 0 && live && fb != 0.0 && ws.phase == WarmSearch::W_FAIL) {
                    nev_first = ws.nev; fb = 0.0; attempt = 1; ws.begin(cprev, dc, l1, 0.0);
                }
            }
        }
        const bool fin = live && lg == 0;            // one lane of the group writes
        ws.nev += nev_first;
        const bool ok = swd_warm_finish(ws, fin, false, betmx, e, chain, nchain, cprev, dc, croot, W);
        if (fin) {
            atomicAdd(&W.stats[1], (unsigned long long)ws.nev); atomicAdd(&W.stats[2], ok ? 1ull : 0ull);
            atomicAdd(&W.stats[26], (unsigned long long)ws.nev); atomicAdd(&W.stats[28], 1ull);
        }
        int nmax = used;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) nmax = max(nmax, __shfl_xor(nmax, off, 64));
        if (lane == 0) atomicAdd(&W.stats[27], (unsigned long long)nmax);      // (per wavefront, as in k_swd_warm: 4 searches here, not 64)
    }
}

// ---------------------------------------------------------------------------------------
// K3c the search WITHOUT a prediction, all periods at once ("swd_cold_scan", round 6, small batches).  A chain the warm search
// declines -- no previous evaluation, a move the first-order model cannot follow, no sign change inside the trust radius: one wild
// configs[0] chain hands 40 % of its steps back for these -- used to go to the reference-semantics search: ~800 dependent rounds
// of evaluations, 2 ms for one chain whatever the batch.  What the later stages need from the warm search is only a root per
// period that is probably the reference's pick: the branch test then walks the reference's own grid for EVERY period of such a
// chain (it is marked `wide`) and the reference-root stage reproduces the reference's refinement, or the chain goes to the
// sequential search after all.  So:
//   k_swd_cold_scan   the secular function of every period on ONE grid per (chain, sequence) -- the model's start value + i dc,
//                     up to the fastest layer + dc, <= COLD_NP points -- lane = grid point, a wavefront = 63 cells, every point
//                     independent; a wavefront that sees a sign change cuts that cell in 63 twice more (lane = point: 1.3e-6
//                     km/s) and leaves the secant point of the last bracket in the period's list of roots;
//   k_swd_cold_pick   a wavefront per (chain, sequence): the reference's scan replayed on those roots, period after period
//                     (getsol, surfdisp96.f:433-479: from the root before - 1.5 dc, upwards if the sign there is the sign below
//                     every root, else downwards, in steps of dc to the first step with an odd number of roots in it) -- no
//                     evaluations, lane = root.
// A pick the roots cannot decide (the scan's clamp at the start value, no root below the fastest layer, more roots than the list
// holds, a not-a-number) fails the chain: it stays on the list.  The pick's last block rewrites the list without the chains that
// came through.
// ---------------------------------------------------------------------------------------
constexpr int COLD_NP = 1024;                 // grid points per period (the first-period walk's limit)
constexpr int COLD_TP = (COLD_NP - 1 + 62) / 63;      // wavefronts (63 cells each, ends shared) per period
constexpr int COLD_NR = 16;                   // roots kept per period
struct SwdCold {
    double* roots;                            // [slot][item of the family][COLD_NR][2]: root, d(secular)/dc there -- in the order they were found;
                                              // slot = position in the hand-back list
    int* nroot;                               // [slot][item] roots found (may exceed COLD_NR; negative: a not-a-number in the row); the pick leaves 0 behind
    int* s0;                                  // [slot][item] sign bit of the function at the start value (below every root)
    int cap;                                  // slots
    int* ticket; int nblocks;                 // blocks of the stage's pick launches (both families), counted as they finish
};
// swd_start_value by a whole wavefront for ONE model of up to 64 layers (every lane the same chain): lane = layer, one load each
// instead of 2 n dependent ones -- the same comparisons in the same order (the slowest layer: the first index that attains the
// minimum), the same gtsolh; betmx2: the fastest S velocity of the f64 layer constants, as k_swd_warm forms it.
__device__ __forceinline__ float swd_start_value_wave(const SwdModel& M, int lane, float& betmx, const double* __restrict__ mdlc, int chain,
                                                      int nchain, float& betmx2) {
    const bool in = lane < M.n;
    const float b = in ? M.Bf(lane) : 0.f, a = in ? M.Af(lane) : 0.f;
    const float c3 = in ? (float)mdlc[((size_t)lane * 6 + 3) * nchain + chain] : -1.e20f;
    const bool sol = b > 0.01f;
    float v = in ? (sol ? b : a) : 1.e20f, bm = in ? b : -1.e20f, b2 = c3;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        v = fminf(v, __shfl_xor(v, off, 64)); bm = fmaxf(bm, __shfl_xor(bm, off, 64)); b2 = fmaxf(b2, __shfl_xor(b2, off, 64));
    }
    betmx = bm; betmx2 = b2;
    // (the serial loop starts from bmn = 1e20 and takes strictly smaller values only)
    const unsigned long long mm = __ballot(in && (sol ? b : a) == v && v < 1.e20f);
    const int jmn = mm ? __ffsll((long long)mm) - 1 : 0;
    const int jsol = mm ? __shfl(sol ? 1 : 0, jmn, 64) : 1;
    const float bj = __shfl(b, jmn, 64), aj = __shfl(a, jmn, 64);
    float cc1 = (jsol == 0) ? v : swd_gtsolh(aj, bj);
    cc1 = 0.95f * cc1; cc1 = 0.90f * cc1;
    return cc1;
}
__device__ __forceinline__ int swd_cold_points(double cc, float bmx) {
    const double dcs = (double)0.005f;
    const double m = floor(((double)bmx + dcs - cc) / dcs) + 2.0;
    return (m >= 2.0 && m <= (double)COLD_NP) ? (int)m : 0;          // 0: the table cannot hold this model's scan
}

template <class F>
__global__ void __launch_bounds__(64)
k_swd_cold_scan(int nchain, int n, SwdSeqs Q, const float* __restrict__ mdl, const double* __restrict__ mdlc, SwdWarm W, SwdCold C)
{
    const int nsel = min(*W.count, C.cap);
    const int lane = threadIdx.x & 63;
    const long tiles = (long)nsel * Q.nper_total * COLD_TP;
    const double dcs = (double)0.005f;
    int nev = 0;
    for (long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int ti = (int)(tile % COLD_TP);
        const long r = tile / COLD_TP;
        const int el = (int)(r % Q.nper_total), pos = (int)(r / Q.nper_total);
        const int chain = W.list[pos];
        const int e = Q.s[0].croot_off + el;
        int seq = 0;
        while (seq + 1 < Q.nseq && e >= Q.s[seq + 1].croot_off) seq++;
        const int k = e - Q.s[seq].croot_off;
        const size_t s = (size_t)n * nchain;
        SwdModel M{mdl + chain, mdl + (F::LOVE && Q.s[seq].alt_vp ? 4 : 1) * s + chain, mdl + 2 * s + chain, mdl + 3 * s + chain, nchain, n};
        float bmx = 0.f, betmx2 = 0.f;
        const double cc = (double)(n <= 64 ? swd_start_value_wave(M, lane, bmx, mdlc, chain, nchain, betmx2) : swd_start_value(M, bmx));
        const int np = swd_cold_points(cc, bmx);
        if (ti * 63 >= np - 1) continue;                                     // (no cell of this wavefront's inside the scan; np = 0: the pick fails the chain)
        const size_t row = (size_t)pos * Q.nper_total + el;
        const int pi = ti * 63 + lane;
        const bool valid = pi < np;
        const double omega = (2.0 * 3.141592653589793) / (Q.s[seq].t[k] * Q.s[seq].scale);
        const double* lc0 = mdlc + chain;
        auto loadL = [&](int m) {
            const double* o = lc0 + (size_t)m * 6 * nchain;
            return SwdLayerC{o[0], o[(size_t)nchain], o[(size_t)2 * nchain], o[(size_t)3 * nchain],
                             o[(size_t)4 * nchain], o[(size_t)5 * nchain]};
        };
        const double x = cc + (double)pi * dcs;
        const double f = valid ? swd_secular_family<F>(n, loadL, omega, x) : 0.0;
        if (valid) nev++;
        const int sg = signbit(f) ? 1 : 0;
        int sgb = __shfl_up(sg, 1, 64);
        if (lane == 0) sgb = sg;
        const unsigned long long mn = __ballot(valid && f != f);
        unsigned long long mf = __ballot(valid && lane > 0 && sg != sgb);           // bit L: a sign change between points L - 1 and L
        if (ti == 0 && lane == 0) C.s0[row] = sg;
        if (mn) { if (lane == 0) atomicAdd(&C.nroot[row], -(1 << 20)); continue; }
        while (mf) {
            const int L = __ffsll((long long)mf) - 1;
            mf &= mf - 1ull;
            double a = __shfl(x, L - 1, 64), b = __shfl(x, L, 64), fa = 0.0, fb = 0.0;
            bool ok = true;
            for (int round = 0; round < 2 && ok; round++) {
                const double xr = lane == 63 ? b : a + (b - a) * ((double)lane / 63.0);
                const double fr = swd_secular_family<F>(n, loadL, omega, xr);
                nev++;
                const int sa = __shfl(signbit(fr) ? 1 : 0, 0, 64);
                const unsigned long long mn2 = __ballot(fr != fr), mc = __ballot((signbit(fr) ? 1 : 0) != sa);
                if (mn2 || !mc) { ok = false; break; }
                const int L2 = __ffsll((long long)mc) - 1;                  // (>= 1: lane 0 has the sign sa)
                a = __shfl(xr, L2 - 1, 64); b = __shfl(xr, L2, 64); fa = __shfl(fr, L2 - 1, 64); fb = __shfl(fr, L2, 64);
            }
            if (lane == 0) {
                if (!ok) atomicAdd(&C.nroot[row], -(1 << 20));               // (a sign change that could not be refined: the row is no use)
                else {
                    const int sl = atomicAdd(&C.nroot[row], 1);
                    if (sl >= 0 && sl < COLD_NR) {
                        double root = a - fa * (b - a) / (fb - fa);
                        if (!(root >= a && root <= b)) root = 0.5 * (a + b);
                        C.roots[(row * COLD_NR + sl) * 2] = root; C.roots[(row * COLD_NR + sl) * 2 + 1] = (fb - fa) / (b - a);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) nev += __shfl_xor(nev, off, 64);
    if (lane == 0 && nev) { atomicAdd(&W.stats[1], (unsigned long long)nev); atomicAdd(&W.stats[33], (unsigned long long)nev); }
}

template <class F>
__global__ void __launch_bounds__(64)
k_swd_cold_pick(int nchain, int n, SwdSeqs Q, const float* __restrict__ mdl, const double* __restrict__ mdlc,
                double* __restrict__ croot, int* __restrict__ sflag, SwdWarm W, SwdCold C)
{
    // a wavefront per (chain, sequence): the sequence's roots into LDS together, the replay by lane 0 (sequential by nature),
    // the results written together
    extern __shared__ double cold_lds[];             // [nper][COLD_NR * 2] roots and slopes, [nper] count, sign at the start value, pick
    const int nsel = min(*W.count, C.cap);
    const int lane = threadIdx.x & 63;
    const double dcs = (double)0.005f;
    for (long it = blockIdx.x; it < (long)nsel * Q.nseq; it += gridDim.x) {
        const int seq = (int)(it % Q.nseq), pos = (int)(it / Q.nseq);
        const int chain = W.list[pos];
        const int nper = Q.s[seq].nper;
        const size_t s = (size_t)n * nchain;
        SwdModel M{mdl + chain, mdl + (F::LOVE && Q.s[seq].alt_vp ? 4 : 1) * s + chain, mdl + 2 * s + chain, mdl + 3 * s + chain, nchain, n};
        float bmx = 0.f, betmx = -1.e20f;
        double cc;
        if (n <= 64) cc = (double)swd_start_value_wave(M, lane, bmx, mdlc, chain, nchain, betmx);
        else {
            cc = (double)swd_start_value(M, bmx);
            for (int m = 0; m < n; m++) betmx = fmaxf(betmx, (float)mdlc[((size_t)m * 6 + 3) * nchain + chain]);   // as k_swd_warm
        }
        const int np = swd_cold_points(cc, bmx);
        const size_t row0 = (size_t)pos * Q.nper_total + (size_t)(Q.s[seq].croot_off - Q.s[0].croot_off);
        double* Rl = cold_lds;
        int* nrl = (int*)(cold_lds + (size_t)nper * COLD_NR * 2);
        int* s0l = nrl + nper;
        int* pkl = s0l + nper;
        __syncthreads();
        for (int i = lane; i < nper * COLD_NR * 2; i += 64) Rl[i] = C.roots[row0 * COLD_NR * 2 + i];
        for (int j = lane; j < nper; j += 64) {
            const int nc = np == 0 ? -1 : C.nroot[row0 + j];
            C.nroot[row0 + j] = 0;                                           // (for the next scan over this row)
            nrl[j] = (nc < 0 || nc > COLD_NR) ? -1 : nc;
            s0l[j] = C.s0[row0 + j];
            pkl[j] = -1;
        }
        __syncthreads();
        for (int j = lane; j < nper; j += 64) {
            // the period's roots in ascending order (they arrive in the order the scan's wavefronts finished)
            double* R = Rl + (size_t)j * COLD_NR * 2;
            const int nc = nrl[j];
            for (int i = 1; i < nc; i++) {
                const double r = R[2 * i], sl = R[2 * i + 1];
                int q = i - 1;
                while (q >= 0 && R[2 * q] > r) { R[2 * q + 2] = R[2 * q]; R[2 * q + 3] = R[2 * q + 1]; q--; }
                R[2 * q + 2] = r; R[2 * q + 3] = sl;
            }
        }
        __syncthreads();
        // statistics 34-39: why a pick failed (34 no table / forced, 35 start point below the start value, 36 unused, 37 no root up
        // to the fastest layer, 38 none down to the start value, 39 not-a-number / an unrefined root / more roots than the list
        // holds / root above the fastest layer)
        int why = (np == 0 || (W.force && W.force[chain])) ? 34 : 0;
        // (sequential in the periods -- every scan starts from the root before -- but within a period lane l takes the l-th root
        // in scan order: 36 periods in ~10 us)
        const int s1 = s0l[0];                                                      // del1st: the sequence's first evaluation
        double rprev = 0.0;
        for (int j = 0; j < nper && !why; j++) {
            const int nr = nrl[j];
            if (nr < 0) { why = 39; break; }                                         // a not-a-number, or more roots than the list holds
            const double* R = Rl + (size_t)j * COLD_NR * 2;
            const int s0 = s0l[j];
            const double c1 = j == 0 ? cc : rprev - 1.5 * dcs;
            if (j > 0 && !(c1 > cc)) { why = 35; break; }
            // sign at the scan's start point: roots below it
            const int nb = __popcll(__ballot(lane < nr && R[2 * (lane < nr ? lane : 0)] < c1));
            const int dir = (j == 0 || ((s0 ^ (nb & 1)) == s1)) ? +1 : -1;
            // The first step of the reference's grid (c1 +- m dc, m = 1, 2, ...) with an odd number of roots in it: root l in scan
            // order, at distance D from the start point, lies in step ceil(D / dc).  (The reference's grid hangs on ITS root of the
            // period before, which is only good to 1e-6 c: a root that close to a grid point may belong to either step, which
            // matters where it changes a step's count from odd to even.  Failing every such chain cost more than it saved -- 7 % of a
            // wild chain's evaluations: the pick goes by this grid, and the branch test and the reference-root stage, which
            // evaluate the reference's points, have the last word as they have for every root.  Three roots in one step: the
            // reference's refinement finds one of them, and so does the stage that repeats it.)
            const int ndir = dir > 0 ? nr - nb : nb;
            const bool in = lane < ndir;
            const int qi = in ? (dir > 0 ? nb + lane : nb - 1 - lane) : 0;
            const double Dl = (double)dir * (R[2 * qi] - c1);
            const int m = in ? (int)ceil(Dl * (1.0 / dcs)) : -1 - lane;            // (in scan order the steps do not decrease: a step's roots are neighbours)
            int mb = __shfl_up(m, 1, 64);
            const bool first = in && (lane == 0 || mb != m);
            const unsigned long long mfirst = __ballot(first);
            // (roots of my step: up to the next lane that starts one)
            const unsigned long long above = lane >= 63 ? 0ull : (mfirst >> (lane + 1));
            const int cnt = (above ? __ffsll((long long)above) : ndir - lane);
            // getsol's limits: upwards the scan ends once it has moved beyond the fastest layer + dc (:477-479), downwards it is
            // clamped at the start value (:463-467, left to the sequential search)
            const bool lim = in && (dir > 0 ? (c1 + (double)(m - 1) * dcs >= (double)bmx + dcs) : (c1 - (double)m * dcs <= cc));
            const unsigned long long me = __ballot(first && ((cnt & 1) || lim));
            if (!me) { why = dir < 0 ? 38 : 37; break; }
            const int L = __ffsll((long long)me) - 1;
            if (__shfl(lim ? 1 : 0, L, 64)) { why = dir > 0 ? 37 : 38; break; }
            const int pick = dir > 0 ? nb + L : nb - 1 - L;
            if (R[2 * pick] > (double)betmx) { why = 39; break; }                   // getsol :483-485: the reference-semantics search decides the flag
            rprev = R[2 * pick];
            if (lane == 0) pkl[j] = pick;
        }
        if (lane == 0) {
        if (why && atomicExch(&W.need[chain], 5) != 5) atomicAdd(&W.stats[why], 1ull);
        // (a chain whose evaluation before this one failed comes through here like any other: its flags are of THAT model)
        if (!why) sflag[(size_t)seq * nchain + chain] = 1;
        }
        __syncthreads();
        for (int j = lane; j < nper; j += 64) {
            const int pick = pkl[j];
            if (pick < 0) continue;
            const double root = Rl[((size_t)j * COLD_NR + pick) * 2], slope = Rl[((size_t)j * COLD_NR + pick) * 2 + 1];
            const size_t o = (size_t)(Q.s[seq].croot_off + j) * nchain + chain;
            croot[o] = (double)(float)root;                                         // surfdisp96.f:302
            if (W.cwarm) W.cwarm[o] = (double)(float)root;
            W.sgn[o] = slope > 0.0 ? 1 : 0;                                         // sign bit of the function just below the root
            W.slope[o] = slope;
            if (W.ferr) W.ferr[o] = 0.0;
            if (j == 0) W.betmx[(size_t)(F::LOVE ? 1 : 0) * nchain + chain] = betmx;
        }
        if (lane == 0 && !why) atomicAdd(&W.stats[2], (unsigned long long)nper);
    }
    // ---- the stage's last block: the list without the chains that came through
    __threadfence();
    int last = 0;
    if (lane == 0) last = atomicAdd(C.ticket, 1) == C.nblocks - 1 ? 1 : 0;
    last = __shfl(last, 0, 64);
    if (!last) return;
    __threadfence();
    if (lane == 0) {
        const int n0 = *W.count;
        int w = 0, rec = 0;
        for (int i = 0; i < n0; i++) {
            const int ch = W.list[i];
            const int nd = atomicAdd(&W.need[ch], 0);
            if (i < C.cap && nd == 1) { W.need[ch] = 0; if (atomicExch(&W.wide[ch], 1) == 0) atomicAdd(&W.stats[13], 1ull); rec++; }
            else { W.need[ch] = 1; W.list[w++] = ch; }
        }
        *W.count = w;
        atomicAdd(&W.stats[0], (unsigned long long)(-(long long)rec));          // (taken back from "chains handed back")
        atomicAdd(&W.stats[32], (unsigned long long)rec);
        *C.ticket = 0;
    }
}

// The branch test of the warm start (WarmSearch, swd_math.hpp): one secular evaluation per item at the point the
// reference's scan of this period would start from -- the continued root of the period before minus 1.5 dc, or the
// model's start value for a sequence's first period (surfdisp96.f:257-276).  mdl: the float32 search model (start value).
template <class F>
__global__ void __launch_bounds__(64)
k_swd_warm_check(int nchain, int n, SwdSeqs Q, const float* __restrict__ mdl, const double* __restrict__ mdlc,
                 const double* __restrict__ croot, SwdWarm W)
{
    const size_t g = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (g >= (size_t)Q.nper_total * nchain) return;
    const int el = (int)(g / nchain), chain = (int)(g - (size_t)el * nchain);
    const int e = Q.s[0].croot_off + el;
    int seq = 0;
    while (seq + 1 < Q.nseq && e >= Q.s[seq + 1].croot_off) seq++;
    const int k = e - Q.s[seq].croot_off;
    if (W.need[chain]) return;                                       // already on its way to the full search
    const int sg = W.sgn[(size_t)e * nchain + chain];
    const double ck = croot[(size_t)e * nchain + chain];
    const double dcs = (double)0.005f;
    const size_t s = (size_t)n * nchain;
    SwdModel M{mdl + chain, mdl + (F::LOVE && Q.s[seq].alt_vp ? 4 : 1) * s + chain, mdl + 2 * s + chain, mdl + 3 * s + chain, nchain, n};
    // REGULAR sequence (normal dispersion at every period: each root above the scan start of its period): one evaluation at
    // the start point.  IRREGULAR (some root has dropped more than 1.5 dc under the previous period's: velocity inversions,
    // crowded spectra -- there the reference's pick among neighbouring modes depends on where its 0.005 km/s grid falls):
    // every period of the sequence walks the reference's own scan grid (getsol :433-479) from its start point, up or down
    // as getsol would, and the first cell with a sign change must be the one that holds the continued root.
    const double* cq = croot + (size_t)Q.s[seq].croot_off * nchain + chain;
    bool irregular = false;
    for (int j = 1; j < Q.s[seq].nper; j++) irregular = irregular || (cq[(size_t)(j - 1) * nchain] - 1.5 * dcs >= cq[(size_t)j * nchain]);
    irregular = irregular || W.wide[chain] != 0;
    float bmx = 0.f;
    double cc = 0.0;
    if (k == 0 || irregular) cc = (double)swd_start_value(M, bmx);
    const double sk = (k == 0) ? cc : cq[(size_t)(k - 1) * nchain] - 1.5 * dcs;
    const double omega = (2.0 * 3.141592653589793) / (Q.s[seq].t[k] * Q.s[seq].scale);
    const double* lc0 = mdlc + chain;
    auto loadL = [&](int m) {
        const double* o = lc0 + (size_t)m * 6 * nchain;
        return SwdLayerC{o[0], o[(size_t)nchain], o[(size_t)2 * nchain], o[(size_t)3 * nchain],
                         o[(size_t)4 * nchain], o[(size_t)5 * nchain]};
    };
    // The reference's scan ends at the first GRID point beyond the root (start + m dc), and above the fastest layer the
    // secular function no longer changes sign there (the half-space term is taken by its absolute value, surfdisp96.f:744,
    // :815): a root within dc of that velocity -- Love waves at long periods -- is found or missed by the grid.  Such a
    // sequence walks the grid as well.
    bool forced = false;
    if (!irregular && sg <= 1 && sk > 0.0 && sk < ck) {
        const double gup = sk + (floor((ck - sk) / dcs) + 1.0) * dcs;
        // (no write to `wide` here: other items of the chain read it in this very launch; the walk reads `irr` instead)
        if (gup > (double)W.betmx[(size_t)(F::LOVE ? 1 : 0) * nchain + chain]) { irregular = true; forced = true; }
    }
    // (a chain marked wide, or a root next to the fastest layer: the whole sequence walks, whatever the window)
    const bool local = irregular && W.walk_window >= 0 && !W.wide[chain] && !forced &&
                       !swd_walk_near(cq, (size_t)nchain, Q.s[seq].nper, k, W.walk_window, dcs) && sk > 0.0 && sk < ck;
    if (irregular && !local) {                                       // -> k_swd_warm_walk, one 16-lane group per item
        if (atomicExch(&W.irr[chain], 1) == 0) W.ilist[atomicAdd(W.icount, 1)] = chain;
        return;
    }
    if (local && cc == 0.0 && k == 0) cc = (double)swd_start_value(M, bmx);
    const bool order = sg > 1 || !(sk > 0.0) || sk == ck;
    const double f = swd_secular_family<F>(n, loadL, omega, order ? ck : sk);
    // (the sign of the sequence's first evaluation, del1st, which the dense walk of its later periods asks for: written by the
    // first period's own walk -- or here, where the first period does not walk)
    if (local && k == 0 && W.sg1) W.sg1[(size_t)((F::LOVE ? 4 : 0) + seq) * nchain + chain] = (unsigned char)(signbit(f) ? 1 : 0);
    int nev = 1;
    const bool bad = order || !(sk < ck) || (signbit(f) ? 1 : 0) != sg;
    if (bad && atomicExch(&W.need[chain], 1) == 0) {
        W.list2[atomicAdd(W.count2, 1)] = chain;
        atomicAdd(&W.stats[0], 1ull);
        // cause: 8 no root / degenerate start point, 9 another root between the scan's start and the continued root,
        // 10 / 11 the same for a sequence's first period
        atomicAdd(&W.stats[(order ? 8 : 9) + (k == 0 ? 2 : 0)], 1ull);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) nev += __shfl_xor(nev, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&W.stats[1], (unsigned long long)nev);
}

// The branch test of IRREGULAR sequences: 16 lanes per (chain, period) item walk the reference's scan grid together --
// lane j evaluates the j-th next grid point of a round, the group's first sign change is found with a ballot.  getsol
// :433-479: direction from the sign at the start point against the sign below every root (del1st: the first evaluation
// of the sequence's first period), steps of dc, abort below the start value or above the fastest layer.  The first cell
// with a sign change must hold the continued root; otherwise the chain goes to the full search.
template <class F, bool FIRST>      // FIRST: the first period of every sequence (a scan of 100-700 cells from the start value): 64 lanes
__global__ void __launch_bounds__(64)   // per item; otherwise the later periods (~2-8 cells): 8 lanes per item, eight items per wavefront
k_swd_warm_walk(int nchain, int n, SwdSeqs Q, const float* __restrict__ mdl, const double* __restrict__ mdlc,
                const double* __restrict__ croot, SwdWarm W)
{
    constexpr int LPI = FIRST ? 64 : 8, IPW = 64 / LPI;      // (a later period's scan is ~2-8 cells long: 8 lanes per item, a second round where needed)
    const int lane = threadIdx.x & 63, sub = lane / LPI, li = lane % LPI;
    const int nsel = *W.icount;
    if (FIRST && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&W.stats[12], (unsigned long long)nsel);     // chains whose sequences walk the grid
    const int per_chain = FIRST ? Q.nseq : Q.nper_total;
    const long total = (long)nsel * per_chain;
    const double dcs = (double)0.005f;
    for (long base = (long)blockIdx.x * IPW; base < total; base += (long)gridDim.x * IPW) {
        const long it = base + sub;
        bool live = it < total;
        const int pos = live ? (int)(it / per_chain) : 0, sub_it = live ? (int)(it - (long)pos * per_chain) : 0;
        const int chain = W.ilist[live ? pos : 0];
        int seq, k;
        if (FIRST) { seq = sub_it; k = 0; }
        else {
            const int e0 = Q.s[0].croot_off + sub_it;
            seq = 0;
            while (seq + 1 < Q.nseq && e0 >= Q.s[seq + 1].croot_off) seq++;
            k = e0 - Q.s[seq].croot_off;
            live = live && k > 0;
        }
        const int e = Q.s[seq].croot_off + k;
        const double* cq = W.walk_roots + (size_t)Q.s[seq].croot_off * nchain + chain;
        bool irregular = false;
        for (int j = 1; j < Q.s[seq].nper; j++) irregular = irregular || (cq[(size_t)(j - 1) * nchain] - 1.5 * dcs >= cq[(size_t)j * nchain]);
        // (with a window: only the periods next to an anomalous pair -- unless the chain is wide, or the sequence is not anomalous
        // at all and walks for another reason, a root next to the fastest layer: then every period does)
        const bool whole = W.wide[chain] != 0 || !irregular;
        irregular = irregular || W.wide[chain] != 0 || W.irr[chain] != 0;      // (irr: set by the branch test, launch before this one)
        live = live && irregular && !W.need[chain] && (whole || swd_walk_near(cq, (size_t)nchain, Q.s[seq].nper, k, W.walk_window, dcs));
        const size_t s = (size_t)n * nchain;
        SwdModel M{mdl + chain, mdl + (F::LOVE && Q.s[seq].alt_vp ? 4 : 1) * s + chain, mdl + 2 * s + chain, mdl + 3 * s + chain, nchain, n};
        float bmx = 0.f;
        const double cc = (double)swd_start_value(M, bmx);
        const int sg = W.sgn[(size_t)e * nchain + chain];
        const double ck = W.walk_roots[(size_t)e * nchain + chain];
        const double sk = (k == 0) ? cc : cq[(size_t)(k - 1) * nchain] - 1.5 * dcs;
        const double omega = (2.0 * 3.141592653589793) / (Q.s[seq].t[k] * Q.s[seq].scale);
        const double om0 = (2.0 * 3.141592653589793) / (Q.s[seq].t[0] * Q.s[seq].scale);
        const double* lc0 = mdlc + chain;
        auto loadL = [&](int m) {
            const double* o = lc0 + (size_t)m * 6 * nchain;
            return SwdLayerC{o[0], o[(size_t)nchain], o[(size_t)2 * nchain], o[(size_t)3 * nchain],
                             o[(size_t)4 * nchain], o[(size_t)5 * nchain]};
        };
        bool bad = live && (sg > 1 || !(sk > 0.0) || sk == ck);
        bool walking = live && !bad;
        // round 0: lane 0 of the group evaluates the start point, lane 1 the sign below every root (del1st).  A sequence's first
        // period (FIRST) has both in one -- its start point IS the sequence's first evaluation and the scan goes up -- and takes
        // it as point 0 of its first round of grid points instead of a round of its own
        constexpr int OFF = FIRST ? 0 : 1;                // index of a round's first point
        double f0 = 0.0;
        if (!FIRST && walking && li < 2) f0 = swd_secular_family<F>(n, loadL, li == 0 ? omega : om0, li == 0 ? sk : cc);
        const int gbase = sub * LPI;
        const double fsk = __shfl(f0, gbase, 64), f1st = __shfl(f0, gbase + 1, 64);
        const int idir = (FIRST || k == 0 || signbit(fsk) == signbit(f1st)) ? +1 : -1;
        int sprev = signbit(fsk) ? 1 : 0;                 // sign at the last point of the round before
        int nev = (!FIRST && walking && li < 2) ? 1 : 0;
        // (a first period's scan starts at the model's start value -- 0.77 x the Rayleigh velocity of the SLOWEST layer -- and may
        // have 2.5 km/s to go on models with one slow layer: 1024 cells; later periods start 1.5 cells below the root before)
        constexpr int MAXR = (FIRST ? 1024 : 400) / LPI;
        for (int round = 0; round < MAXR && __any(walking); round++) {
            const int pidx = round * LPI + li + OFF;                       // 0: the start point itself (FIRST only)
            const double c = pidx == 0 ? sk : sk + (double)idir * (double)pidx * dcs;
            double f = 0.0;
            if (walking) { f = swd_secular_family<F>(n, loadL, omega, (pidx == 0 || c > 1.0e-3) ? c : 1.0e-3); nev++; }
            const int sgn_me = signbit(f) ? 1 : 0;
            int sgn_before = __shfl_up(sgn_me, 1, LPI);
            if (li == 0) sgn_before = (FIRST && round == 0) ? sgn_me : sprev;
            if (FIRST && round == 0 && walking && li == 0 && W.sg1)            // del1st, for the later periods' dense walk
                W.sg1[(size_t)((F::LOVE ? 4 : 0) + seq) * nchain + chain] = (unsigned char)sgn_me;
            // per lane: does the scan END at this point?  a sign change against the point before, or one of getsol's limits
            // once the scan has moved here without one (:463-479; the clamp at clow is handed back to the full search)
            const bool clampd = pidx > 0 && c <= cc;                       // getsol would clamp here instead of evaluating
            const bool change = pidx > 0 && sgn_me != sgn_before && !clampd;
            const bool limit = clampd || (pidx > 0 && c >= (double)bmx + dcs);
            const unsigned long long gmask = LPI == 64 ? ~0ull : (((1ull << (LPI & 63)) - 1ull) << gbase);
            const unsigned long long mend = __ballot(walking && (change || limit)) & gmask;
            const unsigned long long mchange = __ballot(walking && change) & gmask;
            if (walking && mend) {
                const int first = __ffsll((long long)mend) - 1 - gbase;   // the group's first ending point
                const bool by_change = ((mchange >> (gbase + first)) & 1ull) != 0;
                const double cend = sk + (double)idir * (double)(round * LPI + first + OFF) * dcs, cbefore = cend - (double)idir * dcs;
                bad = !by_change || !(fmin(cbefore, cend) < ck && ck < fmax(cbefore, cend));
                walking = false;
            }
            sprev = __shfl(sgn_me, gbase + LPI - 1, 64);
            if (round == MAXR - 1 && walking) { bad = true; walking = false; }
        }
        if (live && bad && li == 0 && atomicExch(&W.need[chain], 1) == 0) {
            W.list2[atomicAdd(W.count2, 1)] = chain;
            atomicAdd(&W.stats[0], 1ull);
            atomicAdd(&W.stats[9 + (k == 0 ? 2 : 0)], 1ull);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) nev += __shfl_xor(nev, off, 64);
        if (lane == 0) atomicAdd(&W.stats[1], (unsigned long long)nev);
    }
}

// The later periods' walk, densely packed (option "swd_walk_dense", default): the continued root says how many grid points the
// reference's scan of an item passes -- m = the index of the first point beyond it -- so nothing is evaluated speculatively:
// a wavefront takes 64 items, a prefix sum of their m + 1 points (start point + grid points 1 .. m) deals the points to the
// lanes round by round, and an item's verdict is the conjunction of its points' (same conditions as k_swd_warm_walk<., false>:
// direction from the sign at the start point against del1st -- stored by the first-period walk --, no sign change and none
// of getsol's limits before point m, a sign change at m, the root strictly inside the last cell).  ~8.5 evaluations per item
// instead of 16 (two rounds of 8 lanes).
template <class F>
__global__ void __launch_bounds__(64)
k_swd_warm_walk_dense(int nchain, int n, SwdSeqs Q, const float* __restrict__ mdl, const double* __restrict__ mdlc,
                      const double* __restrict__ croot, SwdWarm W, int ipb = 64)
{
    // ipb: items per block and trip (<= 64).  Small batches take fewer: a block's trips over its items' points come one after the
    // other, and 64 items of a chain whose sequences all walk are ~450 points -- seven trips where eight blocks make one each
    constexpr int MAXM = 400;                    // (k_swd_warm_walk<., false>'s 50 rounds of 8 points)
    const int lane = threadIdx.x & 63;
    const int nsel = *W.icount;
    const int per_chain = Q.nper_total;
    const long total = (long)nsel * per_chain;
    const double dcs = (double)0.005f;
    __shared__ int s_pre[65], s_bad[64], s_chain[64], s_m[64], s_dir[64], s_s1[64];
    __shared__ double s_sk[64], s_om[64], s_cc[64], s_lim[64];
    for (long base = (long)blockIdx.x * ipb; base < total; base += (long)gridDim.x * ipb) {
        // ---- lane = item
        const long it = base + lane;
        bool live = lane < ipb && it < total;
        const int pos = live ? (int)(it / per_chain) : 0, sub_it = live ? (int)(it - (long)pos * per_chain) : 0;
        const int chain = W.ilist[live ? pos : 0];
        const int e0 = Q.s[0].croot_off + sub_it;
        int seq = 0;
        while (seq + 1 < Q.nseq && e0 >= Q.s[seq + 1].croot_off) seq++;
        const int k = e0 - Q.s[seq].croot_off;
        live = live && k > 0 && W.irr[chain] != 0 && !W.need[chain];
        const int e = Q.s[seq].croot_off + k;
        const double* cq = W.walk_roots + (size_t)Q.s[seq].croot_off * nchain + chain;
        if (live && W.walk_window >= 0 && !W.wide[chain]) {
            bool anom = false;                                         // (as k_swd_warm_walk: a sequence without an anomalous pair walks whole)
            for (int j = 1; j < Q.s[seq].nper; j++) anom = anom || (cq[(size_t)(j - 1) * nchain] - 1.5 * dcs >= cq[(size_t)j * nchain]);
            live = !anom || swd_walk_near(cq, (size_t)nchain, Q.s[seq].nper, k, W.walk_window, dcs);
        }
        const size_t s = (size_t)n * nchain;
        SwdModel M{mdl + chain, mdl + (F::LOVE && Q.s[seq].alt_vp ? 4 : 1) * s + chain, mdl + 2 * s + chain, mdl + 3 * s + chain, nchain, n};
        float bmx = 0.f;
        const double cc = (double)swd_start_value(M, bmx);
        const int sg = W.sgn[(size_t)e * nchain + chain];
        const double ck = W.walk_roots[(size_t)e * nchain + chain];
        const double sk = cq[(size_t)(k > 0 ? k - 1 : 0) * nchain] - 1.5 * dcs;
        bool bad = live && (sg > 1 || !(sk > 0.0) || sk == ck || !(ck == ck));
        const int dir = ck > sk ? +1 : -1;
        int m = 1;
        if (live && !bad) {
            const double m0 = floor(fabs(ck - sk) / dcs);
            if (!(m0 < (double)MAXM)) bad = true;
            else {
                m = max(1, (int)m0);
                while (m <= MAXM && !((double)dir * ((sk + (double)dir * (double)m * dcs) - ck) > 0.0)) m++;
                const double cend = sk + (double)dir * (double)m * dcs, cbefore = cend - (double)dir * dcs;
                if (m > MAXM || !(fmin(cbefore, cend) < ck && ck < fmax(cbefore, cend))) bad = true;
            }
        }
        const int cnt = (live && !bad) ? m + 1 : 0;
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
        const int T = __shfl(incl, 63, 64);
        __syncthreads();                                             // (the round before has read its tables)
        s_pre[lane] = incl - cnt; if (lane == 63) s_pre[64] = incl;
        s_bad[lane] = 0; s_chain[lane] = chain; s_m[lane] = m; s_dir[lane] = dir;
        s_s1[lane] = W.sg1[(size_t)((F::LOVE ? 4 : 0) + seq) * nchain + chain];
        s_sk[lane] = sk; s_om[lane] = (2.0 * 3.141592653589793) / (Q.s[seq].t[k] * Q.s[seq].scale); s_cc[lane] = cc;
        s_lim[lane] = (double)bmx + dcs;
        __syncthreads();
        // ---- lane = grid point
        int sprev = 0, nev = 0;
        for (int q0 = 0; q0 < T; q0 += 64) {
            const int q = q0 + lane;
            const bool act = q < T;
            int lo = 0, hi = 64;                                     // the item j with s_pre[j] <= q < s_pre[j + 1]
#pragma unroll
            for (int b = 0; b < 6; b++) { const int mid = (lo + hi) >> 1; if (s_pre[mid] <= q) lo = mid; else hi = mid; }
            const int j = lo;                                        // (an empty item cannot satisfy the invariant: j has points)
            const int pt = q - s_pre[j], mj = s_m[j], dj = s_dir[j];
            const double skj = s_sk[j], ccj = s_cc[j];
            const double c = pt == 0 ? skj : skj + (double)dj * (double)pt * dcs;
            const double* lc0 = mdlc + s_chain[j];
            auto loadL = [&](int l) {
                const double* o = lc0 + (size_t)l * 6 * nchain;
                return SwdLayerC{o[0], o[(size_t)nchain], o[(size_t)2 * nchain], o[(size_t)3 * nchain],
                                 o[(size_t)4 * nchain], o[(size_t)5 * nchain]};
            };
            double f = 0.0;
            if (act) { f = swd_secular_family<F>(n, loadL, s_om[j], (pt == 0 || c > 1.0e-3) ? c : 1.0e-3); nev++; }
            const int sgn_me = signbit(f) ? 1 : 0;
            int sgn_before = __shfl_up(sgn_me, 1, 64);
            if (lane == 0) sgn_before = sprev;
            bool sbad;
            if (pt == 0) sbad = ((sgn_me == s_s1[j]) ? +1 : -1) != dj;                       // getsol's direction, :433-445
            else if (pt < mj) sbad = sgn_me != sgn_before || c <= ccj || c >= s_lim[j];          // the scan would end before the root's cell
            else sbad = sgn_me == sgn_before || c <= ccj;                                        // ... or pass it
            if (act && sbad) s_bad[j] = 1;
            sprev = __shfl(sgn_me, 63, 64);
        }
        __syncthreads();
        bad = bad || (cnt > 0 && s_bad[lane] != 0);
        if (live && bad && atomicExch(&W.need[chain], 1) == 0) {
            W.list2[atomicAdd(W.count2, 1)] = chain;
            atomicAdd(&W.stats[0], 1ull);
            atomicAdd(&W.stats[9], 1ull);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) nev += __shfl_xor(nev, off, 64);
        if (lane == 0) atomicAdd(&W.stats[1], (unsigned long long)nev);
    }
}

// ---------------------------------------------------------------------------------------
// K3x the reference's own roots from the warm-started ones (ExactGroup, swd_math.hpp): lane = (group of periods of one
// sequence, chain).  A lane walks its group period by period -- origin of the scan grid from the unrounded root of the
// period before, cell from the warm root, the reference's nevill inside -- behind `runup` periods that only serve the
// first origin, and overwrites croot with the float32-rounded results (surfdisp96.f:302).  Chains already on their way
// to the full search are skipped; a lane that cannot do its job (ExactGroup's X_FAIL causes) puts its chain on list3.
// mdl: the float32 search model (start value of the scan of a sequence's first period).
// ---------------------------------------------------------------------------------------
// In rounds (round 6): a wavefront executes what its slowest lane needs -- 47 evaluations where the lanes need 38.5 on average, and
// the stage ends with its slowest wavefront.  With a budget every lane gets that many evaluations; the groups that are unfinished
// then (ExactGroupT::save: the whole machine, in the middle of a period if need be) go to a second launch, k_swd_exact_coop with
// 16 lanes per group.  A machine's sequence of evaluations does not depend on where it runs: the same roots bit for bit.
struct ExactSpill {
    double* d;                    // [EXACT_SPILL_ND][cap]
    unsigned long long* item;     // [cap] the group's global index grp * nchain + chain (~0: not a slot)
    int* count;                   // slots handed out (may exceed cap: the lanes beyond finish in place)
    int cap;
};

// Groups whose run-up did not bring their first origin within the tolerance (ExactGroup's cause 7: a run-up period whose root
// was closed by bisections alone passes the origin's error on undiminished) -- with ONE run-up period 0.3 % of the groups.  They
// are not handed to the sequential search but listed here and done again with a longer run-up (k_swd_exact_coop over the list).
struct ExactRedo { int* list; int* count; int cap; };
__device__ __forceinline__ bool swd_exact_redo(const ExactRedo& R, size_t g) {
    if (!R.list) return false;
    const int sl = atomicAdd(R.count, 1);
    if (sl >= R.cap) return false;               // no room: the chain goes to the sequential search after all
    R.list[sl] = (int)g;
    return true;
}

template <class F>
__global__ void __launch_bounds__(64)
k_swd_exact(int nchain, int n, SwdSeqs Q, int G, int runup, int ngroups, float origin_tol, const float* __restrict__ mdl,
            const double* __restrict__ mdlc, double* __restrict__ croot, SwdWarm W, ExactSpill out, int budget, ExactRedo redo)
{
    const size_t g = (size_t)blockIdx.x * 64 + threadIdx.x;
    bool live = g < (size_t)ngroups * nchain;
    const int grp = live ? (int)(g / nchain) : 0, chain = live ? (int)(g - (size_t)grp * nchain) : 0;
    int seq = 0, gl = grp;
    while (seq + 1 < Q.nseq && gl >= (Q.s[seq].nper + G - 1) / G) { gl -= (Q.s[seq].nper + G - 1) / G; seq++; }
    const int nper = Q.s[seq].nper;
    const int k0 = gl * G, k1 = min(k0 + G, nper), kr = max(0, k0 - runup);
    live = live && k0 < nper && !W.need[chain];
    const size_t s = (size_t)n * nchain;
    const size_t e0 = (size_t)Q.s[seq].croot_off * nchain + chain;
    const double* cw = W.cwarm + e0;
    const double* tper = Q.s[seq].t; const double tscale = Q.s[seq].scale;
    auto approx = [&](int k) { return cw[(size_t)k * nchain]; };
    auto om = [&](int k) { return (2.0 * 3.141592653589793) / (tper[k] * tscale); };
    const double* lc0 = mdlc + chain;
    auto loadL = [&](int m) {
        const double* o = lc0 + (size_t)m * 6 * nchain;
        return SwdLayerC{o[0], o[(size_t)nchain], o[(size_t)2 * nchain], o[(size_t)3 * nchain],
                         o[(size_t)4 * nchain], o[(size_t)5 * nchain]};
    };
    __shared__ double nevtab[24 * 64];         // Neville tables of the wavefront's lanes, one column per lane
    ExactGroup x;
    x.phase = ExactGroup::X_DONE; x.nev = 0; x.cause = 0; x.creq = 1.0; x.omega = 1.0;
    const FmVC vc = fm_vc_load();          // sine / cosine coefficients in vector registers for the whole kernel (cplx.hpp)
    if (live) {
        SwdModel M{mdl + chain, mdl + (F::LOVE && Q.s[seq].alt_vp ? 4 : 1) * s + chain, mdl + 2 * s + chain, mdl + 3 * s + chain, nchain, n};
        float bmx = 0.f;
        const double cc = (double)swd_start_value(M, bmx);
        x.begin(kr, k0, k1, cc, bmx, kr > 0 ? approx(kr - 1) * (1.0 - EXACT_OFFSET) : 0.0, approx, om, nevtab + threadIdx.x, 64, origin_tol);
    }
    int left = budget, used = 0;
    bool spilled = false;
    for (;;) {
        while (__any(x.active() && left > 0)) {
            if (x.active() && left > 0) {
                x.advance(swd_secular_family<F, true>(n, loadL, x.omega, x.creq, &vc));      // (DUAL: this kernel is short of wavefronts)
                left--; used++;
                if (x.phase == ExactGroup::X_DONE) {
                    if (x.wanted()) croot[e0 + (size_t)x.k * nchain] = (double)(float)x.root();       // surfdisp96.f:302
                    x.next(approx, om);                     // (the last period: stays X_DONE)
                }
            }
        }
        // out of budget: the unfinished groups of this wavefront go to the second launch, densely
        const bool unfinished = x.active();
        const unsigned long long um = __ballot(unfinished);
        if (um == 0ull) break;
        int sbase = 0;
        const int nun = __popcll(um);
        if ((threadIdx.x & 63) == 0) sbase = atomicAdd(out.count, nun);
        sbase = __shfl(sbase, 0, 64);
        if (sbase + nun > out.cap) {
            // no room: the slots handed out stay empty, the groups finish here
            if (unfinished) { const int sl = sbase + __popcll(um & ((1ull << (threadIdx.x & 63)) - 1ull)); if (sl < out.cap) out.item[sl] = ~0ull; }
            left = 0x7fffffff;
            continue;
        }
        if (unfinished) {
            const size_t sl = (size_t)sbase + __popcll(um & ((1ull << (threadIdx.x & 63)) - 1ull));
            x.save(out.d + sl, (size_t)out.cap);
            out.item[sl] = (unsigned long long)g;
            spilled = true;
        }
        break;
    }
    if (live && !spilled && x.phase == ExactGroup::X_FAIL && x.cause == 7 && swd_exact_redo(redo, g)) {}
    else if (live && !spilled && x.phase == ExactGroup::X_FAIL && atomicExch(&W.need[chain], 1) == 0) {
        W.list3[atomicAdd(W.count3, 1)] = chain;
        atomicAdd(&W.stats[0], 1ull);
        atomicAdd(&W.stats[14], 1ull);
        if (x.cause >= 1 && x.cause <= 7) atomicAdd(&W.stats[16 + x.cause], 1ull);       // "swd_exact_cause_<k>" (ExactGroup's X_FAIL causes)
    }
    int nev = spilled ? 0 : x.nev, nmax = used;         // (a group passed on is counted where it finishes)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { nev += __shfl_xor(nev, off, 64); nmax = max(nmax, __shfl_xor(nmax, off, 64)); }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&W.stats[15], (unsigned long long)nev);
        // divergence of the stage: evaluations of each wavefront's slowest lane (what the wavefront executes is 64 x this), wavefronts
        atomicAdd(&W.stats[3], (unsigned long long)nmax); atomicAdd(&W.stats[16], 1ull);
    }
}

// The same stage with LG = 16 lanes per group (round 6), for the batches that leave the chip mostly empty (a few hundred chains;
// ONE chain for configs[0]): every lane of a group runs the same machine on the same numbers, per evaluation lane j builds the
// vector-independent numbers of layers j, j + 16, ... into LDS and every lane runs the short vector recurrence over them -- as
// k_swd_warm_coop does for the warm search.  The arithmetic is swd_secular_family<F, true>'s operation for operation (entries of a
// layer, then the raw recurrence top-down with its rescales), so the roots are k_swd_exact's bit for bit, in a third of the time
// per evaluation.  glist / gcount: the groups to do (nullptr: all ngroups * nchain of them); the blocks stride.  n - 1 <= 64.
template <class F>
__global__ void __launch_bounds__(64)
k_swd_exact_coop(int nchain, int n, SwdSeqs Q, int G, int runup, int ngroups, float origin_tol, const float* __restrict__ mdl,
                 const double* __restrict__ mdlc, double* __restrict__ croot, SwdWarm W, const int* __restrict__ glist,
                 const int* __restrict__ gcount, ExactSpill in, ExactRedo redo)
{
    constexpr int LG = 16, NG = 64 / LG, LPL = 4, NENT = F::NENT, NV = F::NV;
    extern __shared__ double xcoop_lds[];        // per group: entries [m][NENT], then the half-space vector [NV]
    __shared__ double nevtab[24 * 64];           // Neville tables, one column per lane (the lanes of a group hold identical ones)
    const int lane = threadIdx.x & 63, grp_l = lane / LG, lg = lane - grp_l * LG;
    double* const ent_g = xcoop_lds + (size_t)grp_l * ((size_t)(n - 1) * NENT + NV);
    double* const hs_g = ent_g + (size_t)(n - 1) * NENT;
    // in.d != nullptr: the groups a first launch of k_swd_exact left unfinished, continued from their saved machines
    const size_t total = in.d ? (size_t)min(*in.count, in.cap) : (glist ? (size_t)*gcount : (size_t)ngroups * nchain);
    const FmVC vc = fm_vc_load();
    for (size_t blk = blockIdx.x; blk * NG < total; blk += gridDim.x) {
        const size_t slot = blk * NG + grp_l;
        bool live = slot < total;
        size_t g = live ? (in.d ? (size_t)in.item[slot] : (glist ? (size_t)glist[slot] : slot)) : 0;
        if (in.d && g == ~0ull) { live = false; g = 0; }
        const int grp = (int)(g / nchain), chain = (int)(g - (size_t)grp * nchain);
        int seq = 0, gl = grp;
        while (seq + 1 < Q.nseq && gl >= (Q.s[seq].nper + G - 1) / G) { gl -= (Q.s[seq].nper + G - 1) / G; seq++; }
        const int nper = Q.s[seq].nper;
        const int k0 = gl * G, k1 = min(k0 + G, nper), kr = max(0, k0 - runup);
        live = live && k0 < nper && (in.d || !W.need[chain]);       // (a saved machine is finished whatever its chain's fate: its count)
        const size_t s = (size_t)n * nchain;
        const size_t e0 = (size_t)Q.s[seq].croot_off * nchain + chain;
        const double* cw = W.cwarm + e0;
        const double* tper = Q.s[seq].t; const double tscale = Q.s[seq].scale;
        auto approx = [&](int k) { return cw[(size_t)k * nchain]; };
        auto om = [&](int k) { return (2.0 * 3.141592653589793) / (tper[k] * tscale); };
        const double* lc0 = mdlc + chain;
        auto loadL = [&](int m) {
            const double* o = lc0 + (size_t)m * 6 * nchain;
            return SwdLayerC{o[0], o[(size_t)nchain], o[(size_t)2 * nchain], o[(size_t)3 * nchain],
                             o[(size_t)4 * nchain], o[(size_t)5 * nchain]};
        };
        ExactGroup x;
        x.phase = ExactGroup::X_DONE; x.nev = 0; x.cause = 0; x.creq = 1.0; x.omega = 1.0;
        x.nv.tab.base = nevtab + threadIdx.x; x.nv.tab.stride = 64;
        if (live && in.d) x.load(in.d + slot, (size_t)in.cap, nevtab + threadIdx.x, 64);
        else if (live) {
            SwdModel M{mdl + chain, mdl + (F::LOVE && Q.s[seq].alt_vp ? 4 : 1) * s + chain, mdl + 2 * s + chain, mdl + 3 * s + chain, nchain, n};
            float bmx = 0.f;
            const double cc = (double)swd_start_value(M, bmx);
            x.begin(kr, k0, k1, cc, bmx, kr > 0 ? approx(kr - 1) * (1.0 - EXACT_OFFSET) : 0.0, approx, om, nevtab + threadIdx.x, 64, origin_tol);
        }
        const SwdLayerC Lhalf = loadL(n - 1);
        SwdLayerC Lmine[LPL];                    // this lane's layers lg, lg + LG, ... stay in registers
#pragma unroll
        for (int q = 0; q < LPL; q++) { const int m = lg + q * LG; Lmine[q] = loadL(m < n - 1 ? m : n - 2); }
        while (__any(x.active())) {
            const bool act = x.active();
            const double omega_raw = x.omega;
            const double omega = omega_raw < 1.0e-4 ? 1.0e-4 : omega_raw, iomega = 1.0 / omega;
            const double wvno = omega_raw / x.creq, wvno2 = wvno * wvno, tt = -2.0 * wvno2;
            if (act) {
#pragma unroll
                for (int q = 0; q < LPL; q++) {
                    const int m = lg + q * LG;
                    if (m < n - 1) {
                        double ent[NENT];
                        F::entries_dual(Lmine[q], wvno, wvno2, omega, iomega, ent, &vc);
#pragma unroll
                        for (int i = 0; i < NENT; i++) ent_g[m * NENT + i] = ent[i];
                    }
                }
                if (lg == LG - 1) {
                    double eh[NV];
                    F::halfspace(Lhalf, wvno, wvno2, omega, iomega, eh);
#pragma unroll
                    for (int j = 0; j < NV; j++) hs_g[j] = eh[j];
                }
            }
            __syncthreads();
            double delta = 0.0;
            if (act) {
                double ev[NV];
#pragma unroll
                for (int j = 0; j < NV; j++) ev[j] = hs_g[j];
                const double* pe = ent_g + (size_t)(n - 2) * NENT;
                for (int m = n - 2; m >= 0; m--, pe -= NENT) {
                    double cur[NENT];
#pragma unroll
                    for (int i = 0; i < NENT; i++) cur[i] = pe[i];
                    F::apply(ev, cur, tt);
                    if ((m & 7) == 0) swd_rescale_pow2_n<NV>(ev);
                }
                delta = swd_finish_n<NV>(ev);
            }
            __syncthreads();
            if (act) {
                x.advance(delta);
                if (x.phase == ExactGroup::X_DONE) {
                    if (x.wanted() && lg == 0) croot[e0 + (size_t)x.k * nchain] = (double)(float)x.root();       // surfdisp96.f:302
                    x.next(approx, om);
                }
            }
        }
        if (live && lg == 0) {
            if (x.phase == ExactGroup::X_FAIL && x.cause == 7 && swd_exact_redo(redo, g)) {}
            else if (x.phase == ExactGroup::X_FAIL && atomicExch(&W.need[chain], 1) == 0) {
                W.list3[atomicAdd(W.count3, 1)] = chain;
                atomicAdd(&W.stats[0], 1ull);
                atomicAdd(&W.stats[14], 1ull);
                if (x.cause >= 1 && x.cause <= 7) atomicAdd(&W.stats[16 + x.cause], 1ull);
            }
            atomicAdd(&W.stats[15], (unsigned long long)x.nev);
        }
    }
}

constexpr int COOP_CL = 1;                       // the consumer builds the deepest finite layer itself
// periods of a search sequence as tables in LDS: omega_k = 2 pi / T_k and 1 / max(omega_k, 1e-4), divided once per block
// instead of once per (lane, period) inside the consumer's serial phase
struct SwdOmegaTab {
    const double* om;
    __device__ __forceinline__ double operator()(int k) const { return (2.0 * 3.141592653589793) / om[k]; }   // period (unused)
    __device__ __forceinline__ double omega(int k) const { return om[k]; }
};
// K3 (cooperative): one 512-thread block = 64 (sequence, chain) items.  Wave 0 is the CONSUMER:
// lane = item, it owns the search state machines and runs the short sequential vector recurrence.
// Waves 1..7 are PRODUCERS: wave p builds, for all 64 items at once (lane = item, so every lane of
// a wave works on the same layer index -> little branch divergence, coalesced constants), the
// vector-independent entries of one layer per chunk of 7 layers into a double-buffered LDS ring,
// one chunk ahead of the consumer.  Nothing is computed twice, and the serial path per secular
// evaluation shrinks to (one layer's entries) + (the 25-FMA recurrence over all layers).
#ifndef RFS_COOP_NC
#define RFS_COOP_NC 1
#endif
constexpr int COOP_NC = RFS_COOP_NC;                // consumer waves (each owns 64/COOP_NC of the block's items)
constexpr int COOP_NP = 8 - COOP_NC;             // producer waves = layers per chunk
// F: the secular function (SwdRayFamily; SwdLoveFamily: 3 entries per layer and a 2-vector, a fifth of the LDS)
template <class F, int NCH>                      // chunks held in registers: (n-1-COOP_CL) <= NCH*COOP_NP
__global__ void __launch_bounds__(512)
k_swd_roots_coop(int nchain, int n, SwdSeqs Q, const float* __restrict__ mdl,
                 const double* __restrict__ mdlc, double* __restrict__ croot, int* __restrict__ sflag,
                 const int* __restrict__ list, const int* __restrict__ count)
{
    // list != nullptr: only the *count chains named there (the chains the warm start handed back, k_swd_warm); the grid
    // is sized for every chain and the blocks beyond the list leave at once
    const int nsel = list ? *count : nchain;
    if ((int)blockIdx.x * 64 >= Q.nseq * nsel) return;
    extern __shared__ double lds[];
    double* req = lds;                           // [4][64]: wvno, wvno2, omega, 1/omega
    int* go = (int*)(lds + 4 * 64);              // go[w]: consumer w has another evaluation
    constexpr int NENT = F::NENT, NV = F::NV;
    double* ent = lds + 4 * 64 + 8;              // [2][COOP_NP][NENT][64]
    double* nev = ent + 2 * COOP_NP * NENT * 64;    // [24][64] Neville tables of the 64 state machines
    double* tper = nev + 24 * 64;                       // [2][nseq][nper_max]: omega_k = 2 pi / T_k, then 1 / max(omega_k, 1e-4)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int IPC = 64 / COOP_NC;            // items per consumer wave
    // block-lane = which of the block's 64 items this thread works on
    const int bl = (wave < COOP_NC) ? wave * IPC + (lane % IPC) : lane;
    int item = blockIdx.x * 64 + bl;
    int seq = item / nsel, chain = item - seq * nsel;
    bool live = seq < Q.nseq;
    if (wave < COOP_NC && lane >= IPC) live = false;     // upper lanes of a consumer wave idle
    if (seq >= Q.nseq) { seq = 0; chain = 0; }
    else if (list) chain = list[chain];
    const int nprod = n - 1 - COOP_CL;           // layers handled by the producers
    const int nch = (nprod + COOP_NP - 1) / COOP_NP;
    int npmax = 0;
    for (int q = 0; q < Q.nseq; q++) npmax = max(npmax, Q.s[q].nper);
    for (int i = threadIdx.x; i < Q.nseq * npmax; i += blockDim.x) {
        int q = i / npmax, k = i - q * npmax;
        const double per = (k < Q.s[q].nper) ? Q.s[q].t[k] * Q.s[q].scale : 1.0;
        const double om = (2.0 * 3.141592653589793) / per;                  // RootSearchT::start_period's TWOPI / T(k)
        tper[i] = om; tper[Q.nseq * npmax + i] = 1.0 / (om < 1.0e-4 ? 1.0e-4 : om);
    }
    if (threadIdx.x < 8) go[threadIdx.x] = 0;
    __syncthreads();
    const double* lc0 = mdlc + chain;
    auto loadL = [&](int m) {
        const double* o = lc0 + (size_t)m * 6 * nchain;
        return SwdLayerC{o[0], o[(size_t)nchain], o[(size_t)2 * nchain], o[(size_t)3 * nchain],
                         o[(size_t)4 * nchain], o[(size_t)5 * nchain]};
    };
    if (wave < COOP_NC) {
        // the consumers are the block's critical path: static priority lets their dependent chains issue first
        __builtin_amdgcn_s_setprio(3);
        const size_t s = (size_t)n * nchain;
        const SwdSeq sq = Q.s[seq];
        SwdModel M{mdl + chain, mdl + (F::LOVE && sq.alt_vp ? 4 : 1) * s + chain, mdl + 2 * s + chain, mdl + 3 * s + chain, nchain, n};
        const SwdOmegaTab T{tper + seq * npmax};
        const double* iom = tper + (Q.nseq + seq) * npmax;
        double* cr = croot + (size_t)sq.croot_off * nchain + chain;
        // agent-scope stores: an eigenfunction launch that runs beside this kernel on the other half of the chip may
        // pick up finished periods (k_swd_eigen, early mode: a root is final once it is non-zero)
        auto out = [&](int k, double v) {
            if (live) __hip_atomic_store(&cr[(size_t)k * nchain], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        RootSearchT<NevTabMem> rs;
        rs.tab.base = nev + bl; rs.tab.stride = 64;
        rs.begin(M, T, sq.nper);
        if (!live) rs.done = 1;
        const SwdLayerC Lhalf = loadL(n - 1), Ldeep = loadL(n - 2);
        for (;;) {
            int more = __any(!rs.done);
            double omega = rs.omega < 1.0e-4 ? 1.0e-4 : rs.omega;
            double wvno = rs.omega / rs.creq, wvno2 = wvno * wvno;
            const double iomega = iom[rs.k < sq.nper ? rs.k : 0];               // 1 / omega from the block's table
            if (lane < IPC) { req[bl] = wvno; req[64 + bl] = wvno2; req[128 + bl] = omega; req[192 + bl] = iomega; }
            if (lane == 0) go[wave] = more;
            __syncthreads();                                     // B0
            int any_more = 0;
#pragma unroll
            for (int w = 0; w < COOP_NC; w++) any_more |= go[w];
            if (!any_more) break;
            double e[NV];
            F::halfspace(Lhalf, wvno, wvno2, omega, iomega, e);
            const double tt = -2.0 * wvno2;
            if (COOP_CL) {                                       // deepest layer: built here, beside chunk 0
                double d15[NENT];
                F::entries(Ldeep, wvno, wvno2, omega, iomega, d15);
                F::apply(e, d15, tt);
            }
            for (int c = 0; c < nch; c++) {
                __syncthreads();                                 // chunk c is in buffer c&1
                const double* eb = ent + (size_t)(c & 1) * COOP_NP * NENT * 64 + bl;
                const int nl = min(COOP_NP, nprod - c * COOP_NP);        // layers in this chunk
                if (nl == COOP_NP) {                             // a full chunk: branch-free code (same-box A/B: search 7.12 -> 6.98 ms)
                    double bA[NENT], bB[NENT];
#pragma unroll
                    for (int q = 0; q < NENT; q++) bA[q] = eb[(size_t)q * 64];
#pragma unroll
                    for (int i = 0; i < COOP_NP; i += 2) {
                        if (i + 1 < COOP_NP) {
#pragma unroll
                            for (int q = 0; q < NENT; q++) bB[q] = eb[(size_t)((i + 1) * NENT + q) * 64];
                        }
                        F::apply(e, bA, tt);
                        if (i + 2 < COOP_NP) {
#pragma unroll
                            for (int q = 0; q < NENT; q++) bA[q] = eb[(size_t)((i + 2) * NENT + q) * 64];
                        }
                        if (i + 1 < COOP_NP) F::apply(e, bB, tt);
                    }
                    F::rescale(e);
                    continue;
                }
                // software pipeline: the LDS reads of layer i+1 are in flight while layer i's 25 FMAs issue
                double bufA[NENT], bufB[NENT];
#pragma unroll
                for (int q = 0; q < NENT; q++) bufA[q] = eb[(size_t)q * 64];
#pragma unroll
                for (int i = 0; i < COOP_NP; i += 2) {
                    if (i + 1 < nl) {
#pragma unroll
                        for (int q = 0; q < NENT; q++) bufB[q] = eb[(size_t)((i + 1) * NENT + q) * 64];
                    }
                    if (i < nl) F::apply(e, bufA, tt);
                    if (i + 2 < nl) {
#pragma unroll
                        for (int q = 0; q < NENT; q++) bufA[q] = eb[(size_t)((i + 2) * NENT + q) * 64];
                    }
                    if (i + 1 < nl) F::apply(e, bufB, tt);
                }
                F::rescale(e);                             // once per chunk
            }
            if (!rs.done) rs.advance(F::finish(e), T, out);
            if (COOP_NC > 1) __syncthreads();                    // B_end: go[] may be rewritten
        }
        if (live) sflag[(size_t)seq * nchain + chain] = rs.flag;
    } else {
        __builtin_amdgcn_s_setprio(2);           // the search is the step's critical path: outrank co-resident RF waves
        const int p = wave - COOP_NC;
        SwdLayerC Lmine[NCH];
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            int m = (n - 2 - COOP_CL) - (c * COOP_NP + p);
            Lmine[c] = loadL(m >= 0 ? m : 0);
        }
        for (;;) {
            __syncthreads();                                     // B0
            int any_more = 0;
#pragma unroll
            for (int w = 0; w < COOP_NC; w++) any_more |= go[w];
            if (!any_more) break;
            double wvno = req[lane], wvno2 = req[64 + lane], omega = req[128 + lane], iomega = req[192 + lane];
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                if (c < nch) {
                    int m = (n - 2 - COOP_CL) - (c * COOP_NP + p);
                    if (m >= 0) {
                        double e15[NENT];
                        F::entries(Lmine[c], wvno, wvno2, omega, iomega, e15);
                        double* eb = ent + (size_t)(c & 1) * COOP_NP * NENT * 64 + (size_t)p * NENT * 64 + lane;
#pragma unroll
                        for (int q = 0; q < NENT; q++) eb[(size_t)q * 64] = e15[q];
                    }
                    __syncthreads();
                }
            }
            if (COOP_NC > 1) __syncthreads();                    // B_end
        }
    }
}

// ---------------------------------------------------------------------------------------
// K4 eigenfunction kernels: lane = (item, chain), item = (sequence, period) of one wave family.
// Writes the phase-velocity kernels of the (flat or flattened) model -- d c / d alpha, beta, rho and
// the interface partial -- and U.  LOVE: slegn96 (alpha kernel = 0); SPH: f64 bldsph model.
// Scratch / outputs are indexed by the GLOBAL item e (both families share croot, krn, ugr, cds).
// ---------------------------------------------------------------------------------------
// WATER: the top layer may be a fluid (vs = 0).  Rayleigh: the fluid branches of sregn96 (swd_math.hpp).  Love: slegn96
// reads array elements it never assigned for such a model (uu(1), exl(1): slegn96.f90:211-222 after :417) and the
// compiled reference returns NaN kernels and NaN group velocities -- so does this (phase velocities are the search's).
// crt != nullptr (the chain's column of crT: dadb, drda dadb per layer): the kernels are stored CHAIN-RULED -- slot 0 = d c / d vs
// = kb + ka dadb + kr drda dadb (model_surf.py:184; sphf: the flattened model's vtp / rtp factors inside), slot 1 = the
// interface partial -- which is all the joint evaluation and the warm start read: half the stores here and half the loads
// there.  crt == nullptr: the four raw classes (B1: libsurf.adjoint_kernel returns them).
template <bool LOVE, bool WATER, class Mdl>
__device__ __forceinline__ void swd_eigen_lane(const Mdl& M, int n, int nchain, size_t ntot, double t, double cp,
                                               double* __restrict__ sc, double* __restrict__ ko,
                                               double* __restrict__ uout, const double* __restrict__ crt,
                                               const double* __restrict__ sphf)
{
    const size_t s = (size_t)n * nchain;
    auto put = [&](int m, double da, double db, double dr, double dh) {
        const size_t lm = (size_t)m * nchain;
        if (crt) {
            if (sphf) { const double vtp = sphf[4 * s + lm], rtp = sphf[6 * s + lm]; da *= vtp; db *= vtp; dr *= rtp; }
            ko[lm] = db + da * crt[lm] + dr * crt[s + lm];
            ko[s + lm] = dh;
        } else { ko[lm] = da; ko[s + lm] = db; ko[2 * s + lm] = dr; ko[3 * s + lm] = dh; }
    };
    const double omega = 2.0 * SR_PI32 / t, wvno = omega / cp;
    if (LOVE && WATER && M.B(0) <= 0.0) {
        const double qnan = __longlong_as_double(0x7ff8000000000000ll);
        for (int m = 0; m < n; m++) put(m, 0.0, qnan, qnan, qnan);
        *uout = qnan; uout[ntot] = 1.0; uout[2 * ntot] = 1.0;
        return;
    }
    if (LOVE) {
        sl_up(M, omega, wvno, [&](int m, double uu, double tt, double exl) {
            double* o = sc + (size_t)m * 6 * ntot;
            o[0] = uu; o[ntot] = tt; o[2 * ntot] = exl;
        });
        SlTotals R = sl_down_energy(M, omega, wvno,
            [&](int m, double& uu, double& tt, double& exl) {
                const double* o = sc + (size_t)m * 6 * ntot;
                uu = o[0]; tt = o[ntot]; exl = o[2 * ntot];
            },
            [&](int m, double db, double dr, double dh) { put(m, 0.0, db, dr, dh); });
        // the kernels stay as emitted: their common factors -- 1 / I1, and `fac` of the interface terms with its flush
        // (slegn96.f90:598-602) -- go to the item's two scale slots and are applied by whoever reads krn (swd_krn)
        uout[ntot] = 1.0 / R.sumi1; uout[2 * ntot] = R.fac;
        *uout = R.ugr;
    } else {
        auto store = [&](int m, const double* cd, double exe) {
            double* o = sc + (size_t)m * 6 * ntot;
#pragma unroll
            for (int i = 0; i < 5; i++) o[(size_t)i * ntot] = cd[i];
            o[(size_t)5 * ntot] = exe;
        };
        sr_up<WATER>(M, omega, wvno, store);
        auto load = [&](int m, double* cd, double& exe) {
            const double* o = sc + (size_t)m * 6 * ntot;
#pragma unroll
            for (int i = 0; i < 5; i++) cd[i] = o[(size_t)i * ntot];
            exe = o[(size_t)5 * ntot];
        };
        auto emit = [&](int m, double da, double db, double dr, double dh) { put(m, da, db, dr, dh); };
        SrTotals R = sr_down_energy<WATER>(M, omega, wvno, load, emit);
        // (1 / (U I0) of energy :1181-1186 and `fac` of getdcdh with its flush :1529-1531: the item's scale slots, swd_krn)
        double sc1 = 1.0 / (R.ugr * R.sumi0), sc2 = R.fac;
        double u = R.ugr;
        if (fabs(u) < 1.0e-36) u = 0.0;                                             // :1703
        // A phase velocity that EQUALS a layer's P or S velocity (both are float32 values: about one (period, chain) item of a
        // bench step) makes that layer's vertical wavenumber exactly zero, and sregn96 divides by it (evalg :719-758, intijr):
        // the reference returns NaN for U and for every kernel of the period -- checked on the compiled reference for every
        // layer incl. the half-space, P and S alike; one float32 step away everything is finite (tests/test_gpu_edges.py) --
        // and its samplers end the trajectory there (hmc.py:177-179).  The sweeps here take the root of a real number and
        // come out finite, so the reference's result is restored by hand: NaN scales turn every kernel of the item into NaN.
        bool hit = false;
        for (int m = 0; m < n; m++) hit = hit || cp == M.A(m) || cp == M.B(m);
        if (hit) { const double qnan = __longlong_as_double(0x7ff8000000000000LL); sc1 = qnan; sc2 = qnan; u = qnan; }
        uout[ntot] = sc1; uout[2 * ntot] = sc2;
        *uout = u;
    }
}

// two waves per SIMD (a few spilled registers) hide the scratch round trip of the two sweeps better than one wave
// with the whole register file: measured 1.27 -> 1.09 ms at 8192 chains x 40 periods x 30 layers (3 waves: 2.4 ms)
// Launch modes.  Plain (early = 0, edone = nullptr): every item of the family.  EARLY (early = 1): items el0 .. el1-1
// while the root search may still be running on other CUs -- the roots are read with agent-scope loads, a wavefront
// whose 64 roots are not all final yet (zero = not written) leaves without a trace, one that is processed marks
// edone[wave].  MOP-UP (early = 0, edone given): everything the early launch did not do.  nchain % 64 == 0 is
// required when edone is used (a wavefront = 64 chains of one item).
// LIST (a template argument so that the two uses are two kernels in a profile): only the chains of a hand-back list
template <bool LOVE, bool SPH, bool LIST = false, bool WATER = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_swd_eigen(int nchain, int n, SwdSeqs Q, size_t ntot, const float* __restrict__ mdl, const double* __restrict__ sph,
            const double* __restrict__ croot, const int* __restrict__ sflag, double* __restrict__ cds,
            double* __restrict__ krn, double* __restrict__ ugr, int el0, int el1, int early, int* __restrict__ edone,
            const int* __restrict__ list, const int* __restrict__ count, const double* __restrict__ crT)
{
    // list != nullptr: only the *count chains named there (the chains a warm-started search handed back to the full
    // search, whose roots have just been rewritten); the grid is sized for every chain
    const int nsel = LIST ? *count : nchain;
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (size_t)(el1 - el0) * nsel) return;
    int el = el0 + (int)(g / nsel), chain = (int)(g - (size_t)(el - el0) * nsel);
    if (LIST) chain = list[chain];
    int e = Q.s[0].croot_off + el;
    int seq = 0;
    while (seq + 1 < Q.nseq && e >= Q.s[seq + 1].croot_off) seq++;
    const size_t gi = (size_t)e * nchain + chain;
    const size_t widx = ((size_t)el * nchain + chain) >> 6;
    double cp;
    if (early) {
        cp = __hip_atomic_load(&croot[gi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!__all(cp != 0.0 && cp == cp)) return;
    } else {
        if (edone && edone[widx]) return;
        if (!sflag[(size_t)seq * nchain + chain]) return;
        cp = croot[gi];
    }
    int k = e - Q.s[seq].croot_off;
    double t = Q.s[seq].t[k] * Q.s[seq].scale;
    const size_t s = (size_t)n * nchain;
    double* ko = krn + (size_t)e * 4 * s + chain;       // [e][q][m][chain]
    if (SPH) {
        SwdModelD M{sph + chain, sph + s + chain, sph + 2 * s + chain, sph + 3 * s + chain, nchain, n};
        swd_eigen_lane<LOVE, WATER>(M, n, nchain, ntot, t, cp, cds + gi, ko, ugr + gi, crT ? crT + chain : nullptr, sph + chain);
    } else {
        SwdModel M{mdl + chain, mdl + s + chain, mdl + 2 * s + chain, mdl + 3 * s + chain, nchain, n};
        swd_eigen_lane<LOVE, WATER>(M, n, nchain, ntot, t, cp, cds + gi, ko, ugr + gi, crT ? crT + chain : nullptr, (const double*)nullptr);
    }
    if (early && (threadIdx.x & 63) == 0) edone[widx] = 1;
}

// The eigenfunction pass over a list of GROUPS of the reference-root stage (ExactSpill's items: grp * nchain + chain): the periods
// of the groups whose machines the stage's second launch finished -- their roots arrive while the pass over all items runs, which
// therefore goes ahead beside that launch and leaves these few per cent to be done again here.  lane = (group of the list, period
// of the group).
template <bool LOVE, bool SPH>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_swd_eigen_groups(int nchain, int n, SwdSeqs Q, size_t ntot, const float* __restrict__ mdl, const double* __restrict__ sph,
                   const double* __restrict__ croot, const int* __restrict__ sflag, double* __restrict__ cds,
                   double* __restrict__ krn, double* __restrict__ ugr, const unsigned long long* __restrict__ gitem,
                   const int* __restrict__ gcount, int gcap, int G, const double* __restrict__ crT)
{
    const int nsel = min(*gcount, gcap);
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < (size_t)nsel * G; g += (size_t)gridDim.x * blockDim.x) {
        const size_t slot = g / G;
        const int kk = (int)(g - slot * G);
        const unsigned long long gid = gitem[slot];
        if (gid == ~0ull) continue;
        const int grp = (int)(gid / (unsigned long long)nchain), chain = (int)(gid - (unsigned long long)grp * nchain);
        int seq = 0, gl = grp;
        while (seq + 1 < Q.nseq && gl >= (Q.s[seq].nper + G - 1) / G) { gl -= (Q.s[seq].nper + G - 1) / G; seq++; }
        const int k = gl * G + kk;
        if (k >= Q.s[seq].nper) continue;
        if (!sflag[(size_t)seq * nchain + chain]) continue;
        const int e = Q.s[seq].croot_off + k;
        const size_t gi = (size_t)e * nchain + chain;
        const double cp = croot[gi];
        const double t = Q.s[seq].t[k] * Q.s[seq].scale;
        const size_t s = (size_t)n * nchain;
        double* ko = krn + (size_t)e * 4 * s + chain;
        if (SPH) {
            SwdModelD M{sph + chain, sph + s + chain, sph + 2 * s + chain, sph + 3 * s + chain, nchain, n};
            swd_eigen_lane<LOVE, false>(M, n, nchain, ntot, t, cp, cds + gi, ko, ugr + gi, crT ? crT + chain : nullptr, sph + chain);
        } else {
            SwdModel M{mdl + chain, mdl + s + chain, mdl + 2 * s + chain, mdl + 3 * s + chain, nchain, n};
            swd_eigen_lane<LOVE, false>(M, n, nchain, ntot, t, cp, cds + gi, ko, ugr + gi, crT ? crT + chain : nullptr, (const double*)nullptr);
        }
    }
}

// Data rows.  Up to four blocks in data order (Rc, Rg, Lc, Lg); a phase block reads the items of its
// sequence directly, a group block combines the three passes T, 1.05 T, 0.95 T as sregnpu / slegnpu do
// (sregn96.f90:1839-1844, slegn96.f90:875-878; first term from the 0.95 T pass: quirk).  With
// sphere != 0 the flat-model kernels are mapped back with vtp / dtp / rtp and tm (sprayl, splove and
// the tail of sregnpu / slegnpu); fwd != 0 selects the phase-velocity conversion of libsurf.forward
// (_flat2sphere) instead of sprayl's.
struct SwdBlk { int type, nrow, off, off1, off2; const double* t; };    // type 0 Rc, 1 Rg, 2 Lc, 3 Lg; off* in items
struct SwdRows { SwdBlk b[4]; int nblk, nswd, sphere, fwd; const double* sphR; const double* sphL; int nitems; };

template <bool SPH>
__device__ __forceinline__ double swd_data_value(const SwdRows& R, const SwdBlk& B, int k, int chain, int nchain,
                                                 const double* __restrict__ croot, const double* __restrict__ ugr)
{
    const bool love = B.type >= 2;
    const size_t i0 = (size_t)(B.off + k) * nchain + chain;
    if (!(B.type & 1)) {
        const double c = croot[i0];
        if (!SPH) return c;
        const double t = B.t[k];
        return R.fwd ? c / f2s_tm(love, t, c) : c / sr_tm(love, c, 2.0 * SR_PI32 / t);
    }
    const double u = ugr[i0];
    if (!SPH) return u;
    return u * sr_tm(love, croot[i0], 2.0 * SR_PI32 / B.t[k]);
}

// One kernel value of item e as the eigenfunction pass defines it: the emitted per-layer factor times the item's scale
// (slot 0: alpha, beta, rho; slot 1: interface, flushed below 1e-38 as the reference does).  ugr: [3][item][chain] = U and
// the two scale slots; s = n * nchain, lm = m * nchain + chain.
// RULED: the chain-ruled storage (swd_eigen_lane): q = 0 d c / d vs, q = 1 the interface partial.
template <bool RULED = false>
__device__ __forceinline__ double swd_krn(const double* __restrict__ krn, const double* __restrict__ ugr, size_t ntot,
                                          int e, int q, size_t s, size_t lm, int nchain, int chain)
{
    const bool iface = RULED ? q == 1 : q == 3;
    const double v = krn[((size_t)e * 4 + q) * s + lm];
    const double sc = ugr[(size_t)(iface ? 2 : 1) * ntot + (size_t)e * nchain + chain];
    const double r = v * sc;
    return (iface && fabs(r) < 1.0e-38) ? 0.0 : r;
}

// kernel q (0 alpha, 1 beta, 2 rho, 3 interface; RULED: 0 vs, 1 interface -- the flattening factors of the vs kernel are
// inside it already) of row k of block B at layer m
template <bool SPH, bool RULED = false>
__device__ __forceinline__ double swd_kernel_value(const SwdRows& R, const SwdBlk& B, int k, int q, int m, int chain,
                                                   int nchain, int n, const double* __restrict__ krn,
                                                   const double* __restrict__ croot, const double* __restrict__ ugr)
{
    const size_t s = (size_t)n * nchain, lm = (size_t)m * nchain + chain;
    const bool love = B.type >= 2;
    const int e0 = B.off + k;
    const size_t ntot = (size_t)R.nitems * nchain;
    const double k0 = swd_krn<RULED>(krn, ugr, ntot, e0, q, s, lm, nchain, chain);
    double fac = 1.0;
    if (SPH && RULED) { if (q == 1) fac = (love ? R.sphL : R.sphR)[(size_t)5 * s + lm]; }
    else if (SPH) fac = (love ? R.sphL : R.sphR)[(size_t)((q < 2) ? 4 : (q == 2 ? 6 : 5)) * s + lm];
    if (!(B.type & 1)) {
        if (!SPH) return k0;
        double tm = sr_tm(love, croot[(size_t)e0 * nchain + chain], 2.0 * SR_PI32 / B.t[k]);
        return k0 * fac / (tm * tm * tm);
    }
    const int e1 = B.off1 + k, e2 = B.off2 + k;
    const double t = B.t[k], t1 = t * (1.0 + 0.05), t2 = t * (1.0 - 0.05);
    const double cg = ugr[(size_t)e0 * nchain + chain], cp = croot[(size_t)e0 * nchain + chain];
    const double uc1 = cg / cp;
    const double k1 = swd_krn<RULED>(krn, ugr, ntot, e1, q, s, lm, nchain, chain);
    const double k2 = swd_krn<RULED>(krn, ugr, ntot, e2, q, s, lm, nchain, chain);
    const double du = uc1 * (2.0 - uc1) * k2 - uc1 * uc1 * t * (k2 - k1) / (t2 - t1);
    if (!SPH) return du;
    const double omega = 2.0 * SR_PI32 / t, tm = sr_tm(love, cp, omega), tm1 = sr_tm1(love, omega, tm);
    return (tm * du + cg * cp * k0 * tm1) * fac;
}

// B1 export: [chain][row][layer] arrays for libsurf.adjoint_kernel (thickness kernel =
// suffix sum of the interface partials, sregn96.f90:1727-1731, 1871-1878); single block.
template <bool SPH>
__global__ void k_swd_export(int nchain, int n, SwdRows R, const double* __restrict__ krn,
                             const double* __restrict__ croot, const double* __restrict__ ugr,
                             double* c, double* dcda, double* dcdb, double* dcdr, double* dcdh)
{
    const SwdBlk B = R.b[0];
    const int nrow = B.nrow;
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nchain * nrow) return;
    int chain = g / nrow, k = g - chain * nrow;
    if (c) c[g] = swd_data_value<SPH>(R, B, k, chain, nchain, croot, ugr);
    if (!dcda) return;
    double suf = 0.0;
    for (int m = n - 1; m >= 0; m--) {
        size_t o = (size_t)g * n + m;
        dcda[o] = swd_kernel_value<SPH>(R, B, k, 0, m, chain, nchain, n, krn, croot, ugr);
        dcdb[o] = swd_kernel_value<SPH>(R, B, k, 1, m, chain, nchain, n, krn, croot, ugr);
        dcdr[o] = swd_kernel_value<SPH>(R, B, k, 2, m, chain, nchain, n, krn, croot, ugr);
        dcdh[o] = suf;
        suf += swd_kernel_value<SPH>(R, B, k, 3, m, chain, nchain, n, krn, croot, ugr);
    }
}

// ---------------------------------------------------------------------------------------
// K5 split in two so that every large array is read coalesced:
//   k_rf_reduce   block = chain, thread = layer : fixed-order sum of the pass-B partials + chain rule -> grad
//   k_swd_combine block = 32 chains x 32 layer slots, half-wave = 32 chains : K.r over the periods (krn is chain-minor),
//                 interface -> thickness suffix sums, weighting, misfit, failure returns
// mode: 0 joint (model_rf_swd_vs_thk.py:66-86), 1 RF only (model_rf.py:137-198), 2 SWD only (model_surf.py:155-228)
// merge != 0 (joint evaluation whose surface-wave part is already in place: k_swd_combine ran first, beside the RF
// sweeps, and wrote its unweighted gradient sums, misfit and flag): the RF part is ADDED, and a chain whose root search failed
// gets the joint failure return (0, zeros, dobs, False: model_rf_swd_vs_thk.py:73-74) -- dsyn here, behind the RF synthetics.
// One chain, all threads of the block (k_rf_reduce; the flow entries run it at the head of k_flow_post instead).
struct RfReduce {
    const double* PG; const double* misfit_rf; const double* cr; const double* dobs;
    int n, npart, rf_only, merge; double wt;
};
__device__ __forceinline__ void rf_reduce_chain(const RfReduce& R, int chain, double* __restrict__ misfit,
                                                double* __restrict__ grad, int* __restrict__ flag,
                                                double* __restrict__ dsyn, int ndata)
{
    const int n = R.n;
    if (R.merge && !flag[chain]) {
        if (dsyn) for (int i = threadIdx.x; i < ndata; i += blockDim.x) dsyn[(size_t)chain * ndata + i] = R.dobs[i];
        return;
    }
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        double dadb = R.cr[((size_t)chain * 2) * n + j], drdadb = R.cr[((size_t)chain * 2 + 1) * n + j];
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        const double* pg = R.PG + (size_t)chain * R.npart * 4 * n;
        for (int p = 0; p < R.npart; p++) {
            s0 += pg[((size_t)p * 4 + 0) * n + j]; s1 += pg[((size_t)p * 4 + 1) * n + j];
            s2 += pg[((size_t)p * 4 + 2) * n + j]; s3 += pg[((size_t)p * 4 + 3) * n + j];
        }
        const double gv = s2 + dadb * s1 + drdadb * s0;                         // model_rf.py:189
        const size_t o = (size_t)chain * 2 * n + j;
        grad[o] = R.merge ? ::fma(R.wt, grad[o], gv) : gv;                      // (k_swd_combine left its unweighted sums)
        grad[o + n] = R.merge ? ::fma(R.wt, grad[o + n], s3) : s3;
    }
    if (R.rf_only && threadIdx.x == 0) { misfit[chain] = R.misfit_rf[chain]; flag[chain] = 1; }
    if (R.merge && threadIdx.x == 0) misfit[chain] = ::fma(R.wt, misfit[chain], R.misfit_rf[chain]);
}
__global__ void __launch_bounds__(MAXL)
k_rf_reduce(int nchain, RfReduce R, double* __restrict__ misfit, double* __restrict__ grad, int* __restrict__ flag,
            double* __restrict__ dsyn, int ndata)
{
    rf_reduce_chain(R, blockIdx.x, misfit, grad, flag, dsyn, ndata);
}

template <bool SPH>
__global__ void __launch_bounds__(1024)
k_swd_combine(int nchain, int n, int mode, int nt, SwdRows R, double wt, const double* __restrict__ misfit_rf,
              const double* krn_g, const double* croot_g,
              const double* ugr_g, const int* __restrict__ sflag, int nseq,
              const double* __restrict__ dobs, double* __restrict__ misfit, double* __restrict__ grad,
              double* __restrict__ dsyn, int* __restrict__ flag, int* __restrict__ wvalid, int rowc, int first, int stage = 0)
{
    // first != 0 (joint evaluation, launched beside the RF sweeps): this kernel writes gradient, misfit and flag FIRST and
    // k_rf_reduce adds the RF part behind it (same sums, the other way round); dsyn of a failed chain is left to it as well
    extern __shared__ double hs[];               // [n][32] interface partial sums (+ [3][nswd][32] row cache when rowc) (+ the staged arrays)
    // stage (ONE chain, round 6): a layer's thread goes through the data rows one after the other and every row costs it a handful
    // of dependent loads -- 60 us for 72 rows, the longest kernel of a one-chain step after the reference-root stage.  The chain's
    // kernels, scales and roots are copied into LDS by all threads first (with one chain the arrays are contiguous as they lie)
    // and the very same loop reads them there: the same operations on the same numbers.
    const double* krn = krn_g; const double* croot = croot_g; const double* ugr = ugr_g;
    if (stage) {
        double* kl = hs + (size_t)(n + (rowc ? 3 * R.nswd : 0)) * 32;
        double* ul = kl + (size_t)R.nitems * 4 * n;
        double* cl = ul + (size_t)3 * R.nitems;
        const int tid = threadIdx.y * blockDim.x + threadIdx.x, nth = blockDim.x * blockDim.y;
        for (int i = tid; i < R.nitems * 4 * n; i += nth) kl[i] = krn_g[i];
        for (int i = tid; i < 3 * R.nitems; i += nth) ul[i] = ugr_g[i];
        for (int i = tid; i < R.nitems; i += nth) cl[i] = croot_g[i];
        __syncthreads();
        krn = kl; ugr = ul; croot = cl;
    }
    // 32 chains x 32 layer slots per block: a wavefront = 32 consecutive chains (256 B segments of the chain-minor
    // arrays) x 2 layer slots, so the grid has nchain/32 blocks -- one per CU at 8192 chains instead of one per two
    const int tx = threadIdx.x & 31, slot = threadIdx.y * 2 + (threadIdx.x >> 5), NS = blockDim.y * 2;
    int chain = blockIdx.x * 32 + tx;
    const bool inb = chain < nchain;
    if (!inb) chain = nchain - 1;
    const int nswd = R.nswd, ndata = nt + nswd;
    const size_t ntot = (size_t)R.nitems * nchain, s = (size_t)n * nchain;
    bool ok = true;
    for (int q = 0; q < nseq; q++) ok = ok && (sflag[(size_t)q * nchain + chain] != 0);
    const double w = (mode == 0) ? wt : 1.0;
    double m_swd = 0.0;
    // Row cache (rowc): what a data row contributes is the same for every layer -- its residual r and, for a phase-velocity
    // row, the item's two kernel scales (swd_krn) -- so each (row, chain) pair is formed ONCE per block, by one of its 32
    // layer slots, and read back from LDS by all of them: rc[0] = r, rc[1] = r x scale(alpha, beta, rho), rc[2] = interface scale
    double* rc = hs + (size_t)n * 32;
    if (rowc) {
        int row = 0;
        for (int b = 0; b < R.nblk; b++) {
            const SwdBlk B = R.b[b];
            for (int k = slot; k < B.nrow; k += NS) {
                double d = ok ? swd_data_value<SPH>(R, B, k, chain, nchain, croot, ugr) : 0.0;
                double r = d - dobs[nt + row + k];
                const size_t gi = (size_t)(B.off + k) * nchain + chain;
                rc[(size_t)(row + k) * 32 + tx] = r;
                rc[(size_t)(nswd + row + k) * 32 + tx] = r * ugr[ntot + gi];
                rc[(size_t)(2 * nswd + row + k) * 32 + tx] = ugr[2 * ntot + gi];
                if (ok && inb && dsyn) dsyn[(size_t)chain * ndata + nt + row + k] = d;
            }
            row += B.nrow;
        }
        __syncthreads();
    }
    for (int j = slot; j < n; j += NS) {
        double gs = 0.0, hj = 0.0;
        const size_t lm = (size_t)j * nchain + chain;
        if (ok) {
            int row = 0;
            for (int b = 0; b < R.nblk; b++) {
                const SwdBlk B = R.b[b];
                const bool plain = rowc && !SPH && !(B.type & 1);          // phase velocities of the flat model: k0 itself
                for (int k = 0; k < B.nrow; k++, row++) {
                    double r;
                    if (rowc) r = rc[(size_t)row * 32 + tx];
                    else {
                        double d = swd_data_value<SPH>(R, B, k, chain, nchain, croot, ugr);
                        r = d - dobs[nt + row];
                        if (j == slot && slot == 0 && inb && dsyn) dsyn[(size_t)chain * ndata + nt + row] = d;
                    }
                    if (j == slot && slot == 0) m_swd += r * r;
                    // (krn: the chain-ruled storage of the eigenfunction pass -- d c / d vs = kb + ka dadb + kr drda dadb of
                    // model_surf.py:184 formed where the kernels were, and the interface partial)
                    if (plain) {
                        const double* kq = krn + (size_t)(B.off + k) * 4 * s + lm;
                        double kh = kq[s] * rc[(size_t)(2 * nswd + row) * 32 + tx];
                        if (fabs(kh) < 1.0e-38) kh = 0.0;
                        gs += rc[(size_t)(nswd + row) * 32 + tx] * kq[0];
                        hj += r * kh;
                        continue;
                    }
                    const double gv = swd_kernel_value<SPH, true>(R, B, k, 0, j, chain, nchain, n, krn, croot, ugr);
                    const double kh = swd_kernel_value<SPH, true>(R, B, k, 1, j, chain, nchain, n, krn, croot, ugr);
                    gs += r * gv;
                    hj += r * kh;
                }
            }
        }
        hs[(size_t)j * 32 + tx] = hj;
        if (inb) {
            size_t o = (size_t)chain * 2 * n + j;
            if (!ok) grad[o] = 0.0;
            else grad[o] = first ? gs : ::fma(w, gs, (mode == 0) ? grad[o] : 0.0);     // (first: unweighted; k_rf_reduce forms the same fma)
        }
    }
    __syncthreads();
    for (int j = slot; j < n; j += NS) {
        double t = 0.0;
        for (int m = j + 1; m < n; m++) t += hs[(size_t)m * 32 + tx];       // interface -> thickness partials
        if (inb) {
            size_t o = (size_t)chain * 2 * n + n + j;
            if (!ok) grad[o] = 0.0;
            else grad[o] = first ? t : ::fma(w, t, (mode == 0) ? grad[o] : 0.0);
        }
    }
    if (!inb) return;
    if (!ok) {
        // failure returns: joint -> (0, zeros, dobs, False); SWD only -> (0, zeros, zeros, False)
        if (dsyn && !first) for (int i = slot; i < ndata; i += NS) dsyn[(size_t)chain * ndata + i] = (mode == 0) ? dobs[i] : 0.0;
        if (slot == 0) { misfit[chain] = 0.0; flag[chain] = 0; if (wvalid) wvalid[chain] = 0; }
        return;
    }
    if (slot == 0) {
        double mr = (mode == 0) ? misfit_rf[chain] : 0.0;
        misfit[chain] = first ? 0.5 * m_swd : ::fma(w, 0.5 * m_swd, mr);
        flag[chain] = 1;
        if (wvalid) wvalid[chain] = 1;          // roots + kernels of this chain can seed the next evaluation (k_swd_warm)
    }
}

// synthetics only (forward of the plugins): SWD part of dsyn + flag
__global__ void k_swd_forward_out(int nchain, int nt, SwdRows R, const double* __restrict__ croot,
                                  const double* __restrict__ ugr, const int* __restrict__ sflag, int nseq,
                                  double* __restrict__ dsyn, int* __restrict__ flag)
{
    int chain = blockIdx.x * blockDim.x + threadIdx.x;
    if (chain >= nchain) return;
    const int nswd = R.nswd, ndata = nt + nswd;
    int ok = 1;
    for (int s = 0; s < nseq; s++) ok = ok && (sflag[(size_t)s * nchain + chain] != 0);
    int row = 0;
    for (int b = 0; b < R.nblk; b++) {
        const SwdBlk B = R.b[b];
        for (int k = 0; k < B.nrow; k++, row++)
            dsyn[(size_t)chain * ndata + nt + row] = !ok ? 0.0 : R.sphere ? swd_data_value<true>(R, B, k, chain, nchain, croot, ugr)
                                                                         : swd_data_value<false>(R, B, k, chain, nchain, croot, ugr);
    }
    flag[chain] = ok;
}

// ---------------------------------------------------------------------------------------
// K6 leapfrog pieces (pyhmc/hmc.py:121-201).  One thread per (chain, component) / per chain.
// state: 1 = running, 0 = failed (reference returns early), frozen chains keep their result.
// ---------------------------------------------------------------------------------------
// minv (may be null = identity): diagonal inverse mass, x' = M^-1 p and K = p.M^-1 p / 2 (the reference carries an
// identity `invert_Mass` that only enters its kinetic energy, pyhmc/hmc.py:48,102-108).
__global__ void k_leap_begin(int nchain, int nx, int ndata, const double* minv, const double* x0, const double* p0, const double* dt,
                             const int* L, int Lmax, const double* U, const double* grad, const double* dsyn, const int* flag,
                             double* x, double* p, double* Ucur, double* Hcur, double* Unew, double* Hnew,
                             double* dsyn_cur, double* dsyn_new, int* ok)
{
    __shared__ double red[4];
    __shared__ int bad;
    int chain = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) bad = 0;
    __syncthreads();
    double k = 0.0;
    int mybad = 0;
    for (int i = tid; i < ndata; i += blockDim.x) {
        double d = dsyn[(size_t)chain * ndata + i];
        if (d != d) mybad = 1;
        dsyn_cur[(size_t)chain * ndata + i] = d; dsyn_new[(size_t)chain * ndata + i] = d;
    }
    for (int i = tid; i < nx; i += blockDim.x) {
        double pv = p0[(size_t)chain * nx + i];
        k += pv * pv * (minv ? minv[i] : 1.0);
        x[(size_t)chain * nx + i] = x0[(size_t)chain * nx + i];
        p[(size_t)chain * nx + i] = pv - dt[chain] * grad[(size_t)chain * nx + i] * 0.5;   // hmc.py:164
    }
    if (mybad) atomicOr(&bad, 1);
    k = wave_sum(k);
    if ((tid & 63) == 0) red[tid >> 6] = k;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) s += red[i];
        // hmc.py:155-156; a trajectory length outside [1, Lmax] would never reach its last step: such a chain is
        // reported failed instead of returning stale results
        int good = flag[chain] && !bad && L[chain] >= 1 && L[chain] <= Lmax;
        Ucur[chain] = U[chain]; Unew[chain] = U[chain];
        Hcur[chain] = 0.5 * s + U[chain];                                 // hmc.py:153,157
        Hnew[chain] = __longlong_as_double(0x7ff0000000000000LL);         // +inf until the trajectory completes
        ok[chain] = good;
    }
}

// x += dt p ; mirror reflection at the bounds (hmc.py:121-137, 166-169); only chains still
// inside their trajectory (step < L) and not failed move.
__global__ void k_leap_drift(int nchain, int nx, int step, const double* minv, const double* dt, const int* L,
                             const double* bounds, double* x, double* p, int* ok, int* wforce)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nchain * nx) return;
    int chain = g / nx, i = g - chain * nx;
    if (wforce && i == 0) wforce[chain] = step == L[chain] - 1;      // option swd_exact_final: the end model by the full search
    if (!ok[chain] || step >= L[chain]) return;
    double xv = x[g] + dt[chain] * (p[g] * (minv ? minv[i] : 1.0)), pv = p[g];
    flow_mirror(xv, pv, bounds[2 * i], bounds[2 * i + 1]);
    x[g] = xv; p[g] = pv;
}

// kick after the evaluation at the new x (hmc.py:170-183); finishes the trajectory at step L-1
__global__ void k_leap_kick(int nchain, int nx, int ndata, int step, const double* minv, const double* dt, const int* L,
                            const double* x, const double* U, const double* grad, const double* dsyn,
                            const int* flag, double* p, double* Unew, double* Hnew, double* dsyn_new,
                            double* xnew, int* ok)
{
    __shared__ double red[4];
    __shared__ int bad;
    int chain = blockIdx.x, tid = threadIdx.x;
    if (!ok[chain] || step >= L[chain]) return;
    if (tid == 0) bad = 0;
    __syncthreads();
    int mybad = 0;
    for (int i = tid; i < nx; i += blockDim.x) {
        double xv = x[(size_t)chain * nx + i], gv = grad[(size_t)chain * nx + i];
        if (xv != xv || gv != gv) mybad = 1;
    }
    for (int i = tid; i < ndata; i += blockDim.x) {
        double d = dsyn[(size_t)chain * ndata + i];
        if (d != d) mybad = 1;
    }
    if (mybad) atomicOr(&bad, 1);
    __syncthreads();
    int fail = bad || !flag[chain];
    if (fail) { if (tid == 0) ok[chain] = 0; return; }
    bool last = (step == L[chain] - 1);
    double k = 0.0;
    for (int i = tid; i < nx; i += blockDim.x) {
        double pv = p[(size_t)chain * nx + i] - dt[chain] * grad[(size_t)chain * nx + i] * (last ? 0.5 : 1.0);
        p[(size_t)chain * nx + i] = pv;
        k += pv * pv * (minv ? minv[i] : 1.0);
        if (last) xnew[(size_t)chain * nx + i] = x[(size_t)chain * nx + i];
    }
    if (last) for (int i = tid; i < ndata; i += blockDim.x)
        dsyn_new[(size_t)chain * ndata + i] = dsyn[(size_t)chain * ndata + i];
    k = wave_sum(k);
    if ((tid & 63) == 0) red[tid >> 6] = k;
    __syncthreads();
    if (tid == 0 && last) {
        double s = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) s += red[i];
        Unew[chain] = U[chain];
        Hnew[chain] = 0.5 * s + U[chain];                                 // hmc.py:186-190
    }
}

// ---------------------------------------------------------------------------------------
// K7 continuous-flow leapfrog: every chain sits at its own point of its own trajectory, one call = one
// evaluation per chain.  rem[chain] = leapfrog steps still to do (-1: idle), fresh[chain] = 1: the trajectory starts
// with this call (x = start model, p = drawn momentum).  Same arithmetic as k_leap_begin / drift / kick.
// ---------------------------------------------------------------------------------------
// Device-side restart of a completed trajectory (rfs_flow_next, include/rfsurf.h): all pointers device, have == nullptr = off
struct FlowNext {
    int* have; const double* u; const double* p; const int* rem;
    double* xstart; double* res_x; double* res_val; double* res_dsyn;
    double* gsave; int* kick;        // deferred first half kick (rem == nullptr): gradient at the start model, flag
    // warm start of the root search (library-internal, nullptr = off): the roots of a trajectory's START model are kept
    // aside; a rejected trajectory goes back to that model, and with it go its roots -- the evaluation that follows then
    // continues from "the same model" instead of from the end model a whole trajectory away
    double* croot; double* crs; double* xw; int nitems;
};
// Records of the chains that completed a trajectory in a flow step (rfs_flow_records, include/rfsurf.h): what the caller's books
// need of every such chain, packed densely into a RING in memory the HOST can read (pinned, mapped into the device): the device
// writes a few hundred records of (8 + 2n [+ ndata]) doubles per step over the link, and the host reads them behind the step's
// event instead of copying flags down, indices up and gathered rows down again (a dozen copies and index kernels per step).
//   rec[0] chain, [1] done code (1 waits for the caller, 2 rejected / 3 accepted and restarted on the device), [2] ok,
//   [3] Ucur, [4] Hcur, [5] Hnew, [6] Unew, [7] the call's stamp, [8 .. 8 + nx) end model, then the synthetics if asked for
// A slot is the next value of ONE device counter that only ever goes up (modulo the ring): only the chains that completed take
// one (~8 % of the blocks), nothing is reset between steps and no block has to know that it is the last -- a per-step count
// published by the step's last block cost 8192 atomics on one address, 70 us at the very end of every step.
struct FlowRec {
    double* buf;            // [cap][stride]
    unsigned long long* count;
    int cap, stride, want_dsyn;
    double stamp;
};
__device__ __forceinline__ void flow_post_body(int nchain, int nx, int ndata, const double* minv, const double* dt, double* x, double* U,
                            double* grad, double* dsyn, int* flag, double* p, int* rem, int* fresh,
                            double* Ucur, double* Hcur, double* Unew, double* Hnew, double* dsyn_cur,
                            double* dsyn_new, int* ok, int* done, FlowNext nx_, unsigned long long* fcount, RfReduce rr,
                            const int* need_cur, int* pend, int slot1, unsigned ready);
// rr.PG != nullptr: the RF reduction of this chain's evaluation is still open (joint_eval left it to this kernel)
__global__ void k_flow_post(int nchain, int nx, int ndata, const double* minv, const double* dt, double* x, double* U,
                            double* grad, double* dsyn, int* flag, double* p, int* rem, int* fresh,
                            double* Ucur, double* Hcur, double* Unew, double* Hnew, double* dsyn_cur,
                            double* dsyn_new, int* ok, int* done, FlowNext nx_, unsigned long long* fcount, RfReduce rr,
                            const int* need_cur, int* pend, int slot1, unsigned ready, FlowRec rec)
{
    flow_post_body(nchain, nx, ndata, minv, dt, x, U, grad, dsyn, flag, p, rem, fresh, Ucur, Hcur, Unew, Hnew, dsyn_cur, dsyn_new,
                   ok, done, nx_, fcount, rr, need_cur, pend, slot1, ready);
    if (!rec.buf) return;
    __shared__ unsigned long long s_slot;
    const int chain = blockIdx.x, tid = threadIdx.x;
    __syncthreads();                                            // (the body's stores of this block are complete)
    const int dn = ((volatile int*)done)[chain];
    if (dn == 0) return;                                        // (block-uniform)
    if (tid == 0) s_slot = atomicAdd(rec.count, 1ull);
    __syncthreads();
    double* r = rec.buf + (size_t)(s_slot % (unsigned long long)rec.cap) * rec.stride;
    const bool parked = dn >= 2;                                // restarted on the device: the results wait in res_*
    const double* xs = parked ? nx_.res_x + (size_t)chain * nx : x + (size_t)chain * nx;
    for (int i = tid; i < nx; i += blockDim.x) r[8 + i] = xs[i];
    if (rec.want_dsyn) {
        // (a restarted chain's synthetics at the end model: the evaluation's own row, which k_flow_post parks in res_dsyn)
        const double* ds = parked ? dsyn + (size_t)chain * ndata : dsyn_new + (size_t)chain * ndata;
        for (int i = tid; i < ndata; i += blockDim.x) r[8 + nx + i] = ds[i];
    }
    if (tid == 0) {
        const double* rv = nx_.res_val + (size_t)chain * 4;
        r[0] = (double)chain; r[1] = (double)dn; r[2] = parked ? 1.0 : (double)((volatile int*)ok)[chain];
        r[3] = parked ? rv[0] : ((volatile double*)Ucur)[chain]; r[4] = parked ? rv[1] : ((volatile double*)Hcur)[chain];
        r[5] = parked ? rv[2] : ((volatile double*)Hnew)[chain]; r[6] = parked ? rv[3] : ((volatile double*)Unew)[chain];
        r[7] = rec.stamp;
    }
}

__device__ __forceinline__ void flow_post_body(int nchain, int nx, int ndata, const double* minv, const double* dt, double* x, double* U,
                            double* grad, double* dsyn, int* flag, double* p, int* rem, int* fresh,
                            double* Ucur, double* Hcur, double* Unew, double* Hnew, double* dsyn_cur,
                            double* dsyn_new, int* ok, int* done, FlowNext nx_, unsigned long long* fcount, RfReduce rr,
                            const int* need_cur, int* pend, int slot1, unsigned ready)
{
    __shared__ double red[4];
    __shared__ int bad;
    __shared__ int acc_s;
    const int chain = blockIdx.x, tid = threadIdx.x;
    const int fr = fresh[chain], rm = rem[chain];
    if (tid == 0) done[chain] = 0;
    const bool idle = !fr && (rm <= 0 || !ok[chain]);
    // A chain this step handed back to the full search (rfs_set_option flow_async_handback): its search runs in the
    // background, beside the next step; here it is left exactly as it is -- drifted, not evaluated, not kicked -- and the
    // next call completes the step from the search's roots (no drift, no warm start: `pend`).
    // pend[chain] = 1 + the slot of flags / lists its search was recorded under (0: none); `ready`: the slots whose searches
    // were complete when this call started -- a chain whose search is still running sits this call out as well.
    const int pd = pend ? pend[chain] : 0;
    const bool waiting = pd > 0 && !((ready >> (pd - 1)) & 1u);
    const bool handed = !idle && !waiting && need_cur && need_cur[chain] == 1;
    if (tid == 0 && pend) pend[chain] = handed ? slot1 : (waiting ? pd : 0);
    if (idle || handed || waiting) return;                      // (block-uniform)
    if (tid == 0 && fcount) atomicAdd(&fcount[chain & 63], 1ull);   // statistic "flow_chain_steps" (64 slots: 8192 atomics on ONE address cost 70 us)
    if (rr.PG) rf_reduce_chain(rr, chain, U, grad, flag, dsyn, ndata);
    if (tid == 0) bad = 0;
    __syncthreads();
    int mybad = 0;
    for (int i = tid; i < ndata; i += blockDim.x) {
        double d = dsyn[(size_t)chain * ndata + i];
        if (d != d) mybad = 1;
    }
    if (!fr) for (int i = tid; i < nx; i += blockDim.x) {
        double xv = x[(size_t)chain * nx + i], gv = grad[(size_t)chain * nx + i];
        if (xv != xv || gv != gv) mybad = 1;
    }
    if (mybad) atomicOr(&bad, 1);
    __syncthreads();
    const int fail = bad || !flag[chain];
    if (fr) {                                                   // hmc.py:150-164
        double k = 0.0;
        const bool defer = nx_.kick != nullptr;                 // the step size may still be on its way: kick in the next call
        for (int i = tid; i < nx; i += blockDim.x) {
            double pv = p[(size_t)chain * nx + i];
            k += pv * pv * (minv ? minv[i] : 1.0);
            if (defer) nx_.gsave[(size_t)chain * nx + i] = grad[(size_t)chain * nx + i];
            else p[(size_t)chain * nx + i] = pv - dt[chain] * grad[(size_t)chain * nx + i] * 0.5;
        }
        for (int i = tid; i < ndata; i += blockDim.x) {
            double d = dsyn[(size_t)chain * ndata + i];
            dsyn_cur[(size_t)chain * ndata + i] = d; dsyn_new[(size_t)chain * ndata + i] = d;
        }
        if (nx_.have)                                           // the model this trajectory starts from (kept for a rejection)
            for (int i = tid; i < nx; i += blockDim.x) nx_.xstart[(size_t)chain * nx + i] = x[(size_t)chain * nx + i];
        if (nx_.have && nx_.crs)                                // ... and its roots
            for (int e = tid; e < nx_.nitems; e += blockDim.x) nx_.crs[(size_t)e * nchain + chain] = nx_.croot[(size_t)e * nchain + chain];
        k = wave_sum(k);
        if ((tid & 63) == 0) red[tid >> 6] = k;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int i = 0; i < (int)(blockDim.x >> 6); i++) s += red[i];
            Ucur[chain] = U[chain]; Unew[chain] = U[chain];
            Hcur[chain] = 0.5 * s + U[chain];
            Hnew[chain] = __longlong_as_double(0x7ff0000000000000LL);          // +inf until the trajectory completes
            ok[chain] = !fail;
            fresh[chain] = 0;
            if (defer) nx_.kick[chain] = !fail;
            if (fail) { rem[chain] = -1; done[chain] = 1; }
        }
        return;
    }
    if (tid == 0 && nx_.kick) nx_.kick[chain] = 0;              // the drift (k_prep_joint) has applied the deferred half kick (p holds it)
    if (fail) { if (tid == 0) { ok[chain] = 0; rem[chain] = -1; done[chain] = 1; } return; }
    const bool last = (rm == 1);                                // hmc.py:170-190
    double k = 0.0;
    for (int i = tid; i < nx; i += blockDim.x) {
        double pv = p[(size_t)chain * nx + i] - dt[chain] * grad[(size_t)chain * nx + i] * (last ? 0.5 : 1.0);
        p[(size_t)chain * nx + i] = pv;
        k += pv * pv * (minv ? minv[i] : 1.0);
    }
    if (last) for (int i = tid; i < ndata; i += blockDim.x)
        dsyn_new[(size_t)chain * ndata + i] = dsyn[(size_t)chain * ndata + i];
    k = wave_sum(k);
    if ((tid & 63) == 0) red[tid >> 6] = k;
    __syncthreads();
    const bool auto_next = last && nx_.have && nx_.have[chain];   // block-uniform
    if (tid == 0) {
        if (last) {
            double s = 0.0;
            for (int i = 0; i < (int)(blockDim.x >> 6); i++) s += red[i];
            const double hn = 0.5 * s + U[chain];
            Unew[chain] = U[chain];
            Hnew[chain] = hn;
            if (auto_next) {                                    // hmc.py:192-198 on the device, with the caller's draw
                // (the device's exp against numpy's on the host path: the two can differ in the last bit, so a decision
                // could differ only where u falls within one ulp of the threshold -- ~1e-16 per trajectory)
                const int acc = nx_.u[chain] < exp(-(hn - Hcur[chain]));
                acc_s = acc;
                double* rv = nx_.res_val + (size_t)chain * 4;
                rv[0] = Ucur[chain]; rv[1] = Hcur[chain]; rv[2] = hn; rv[3] = U[chain];
                rem[chain] = nx_.rem ? nx_.rem[chain] : (1 << 30);      // deferred: the caller sets the length before the first step
                fresh[chain] = 1; done[chain] = 2 + acc;
            } else {
                rem[chain] = -1; done[chain] = 1;
            }
        } else {
            rem[chain] = rm - 1;
        }
    }
    if (auto_next) {
        __syncthreads();
        const int acc = acc_s;
        for (int i = tid; i < nx; i += blockDim.x) {
            const size_t o = (size_t)chain * nx + i;
            nx_.res_x[o] = x[o];                                // end model of the trajectory, whatever its fate
            if (!acc) x[o] = nx_.xstart[o];                     // rejected: back to the start model
            p[o] = nx_.p[o];                                    // momentum of the next trajectory
            if (!acc && nx_.crs) nx_.xw[o] = nx_.xstart[o];     // ... which is then also "the previous model" of the root search
        }
        if (!acc && nx_.crs)
            for (int e = tid; e < nx_.nitems; e += blockDim.x) nx_.croot[(size_t)e * nchain + chain] = nx_.crs[(size_t)e * nchain + chain];
        if (nx_.res_dsyn)
            for (int i = tid; i < ndata; i += blockDim.x) nx_.res_dsyn[(size_t)chain * ndata + i] = dsyn[(size_t)chain * ndata + i];
        __syncthreads();
        if (tid == 0) nx_.have[chain] = 0;
    }
}

// rfs_flow_restart: what the caller does to the chains that go through the host between two flow steps (a trajectory that
// ended in a failed evaluation, a run without device-side restarts), in ONE launch: row idx1[i] of x <- xkeep[i]; for the
// chains idx2 that start another trajectory p <- pnew, rem <- L, dt (optional), fresh = ok = 1 (pnew == nullptr: chains already
// under way whose length and step size follow late -- rem and dt only); deposits of idx3 withdrawn.
// block = one listed chain.
__global__ void k_flow_restart(int nx, int n1, int n2, int n3, const int* __restrict__ idx1, const double* __restrict__ xkeep,
                               const int* __restrict__ idx2, const double* __restrict__ pnew, const int* __restrict__ remnew,
                               const double* __restrict__ dtnew, const int* __restrict__ idx3, double* __restrict__ x,
                               double* __restrict__ p, int* __restrict__ rem, double* __restrict__ dt, int* __restrict__ fresh,
                               int* __restrict__ ok, int* __restrict__ nxt_have)
{
    const int b = blockIdx.x;
    if (b < n1) {
        const size_t o = (size_t)idx1[b] * nx, i = (size_t)b * nx;
        for (int j = threadIdx.x; j < nx; j += blockDim.x) x[o + j] = xkeep[i + j];
    } else if (b < n1 + n2) {
        const int r = b - n1, c = idx2[r];
        const size_t o = (size_t)c * nx, i = (size_t)r * nx;
        if (pnew) for (int j = threadIdx.x; j < nx; j += blockDim.x) p[o + j] = pnew[i + j];
        if (threadIdx.x == 0) { rem[c] = remnew[r]; if (dtnew) dt[c] = dtnew[r]; if (pnew) { fresh[c] = 1; ok[c] = 1; } }
    } else if (b < n1 + n2 + n3) {
        if (threadIdx.x == 0 && nxt_have) nxt_have[idx3[b - n1 - n2]] = 0;
    }
}

// rfs_flow_deposit: the deposits of rfs_flow_next for the listed chains in one launch (block = one listed chain): the
// acceptance draw, the next trajectory's momentum and length, then the flag.
__global__ void k_flow_deposit(int nx, int n, const int* __restrict__ idx, const double* __restrict__ u, const double* __restrict__ pn,
                               const int* __restrict__ remn, int* __restrict__ have, double* __restrict__ nu, double* __restrict__ np_,
                               int* __restrict__ nrem)
{
    const int b = blockIdx.x;
    if (b >= n) return;
    const int c = idx[b];
    for (int j = threadIdx.x; j < nx; j += blockDim.x) np_[(size_t)c * nx + j] = pn[(size_t)b * nx + j];
    if (threadIdx.x == 0) { nu[c] = u[b]; if (remn && nrem) nrem[c] = remn[b]; }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) have[c] = 1;
}

}  // namespace rfs

