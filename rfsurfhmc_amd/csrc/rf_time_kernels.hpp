// Time-domain receiver functions (method = "time"): iterative spike deconvolution.
//
// What it reproduces (reference file:line under src/RF):
//   cal_rf_time RFModule.f90:144-191, cal_rf_par_time_all :76-142 (cal_rf_par_time :11-74),
//   deconit deconit.f90:135-197 with gauss_filter :15-32, apply_gaussian :34-52, shift_data :54-72,
//   mycorrelate :100-116, myconvolve :118-133.
//
// How it is re-designed.  The reference runs, per trace (1 + 4*nlayer of them per evaluation) and per
// iteration (up to 200), eight length-nft FFTs to rebuild the residual and its correlation with the
// filtered vertical component.  Every one of those quantities is LINEAR in the spike train P, and a new
// spike of amplitude a at lag i changes
//     rflt(t)  by  -dt a wflt(t - i)                       (gauss(P) * wcopy = P * wflt, wflt = gauss(w))
//     cuw(j)   by  -c Aw(j - i) / Aw(0),  c = cuw(i)       (Aw = circular autocorrelation of wflt)
//     sum(rflt^2) by -a c                                   (a = c / (Aw(0) dt^2))
// so the loop needs NO FFT at all: one wavefront per trace keeps cuw(1:nft/2) and P in registers, finds the
// arg-max with a butterfly, and applies the shifted autocorrelation.  What remains of the FFT work is one
// inverse transform per trace for the initial correlation cuw0 = dt irfft(G^2 u^ conj(w^)), taken straight
// from the spectra (the reference's irfft -> zero-pad -> rfft round trip is the identity), and three per
// chain (Aw for both denominators, cuw0 of the forward trace).  The final gauss + shift of P is a sparse
// sum of shifted copies of one precomputed pulse gsh = irfft(G e^{-i w tshift}); for the misfit gradient
// the traces are never formed: sum_t k(t) r(t) = sum_spikes a * Cres(lag), Cres = correlation of the residual
// with gsh, once per chain.
#pragma once
#include "rfsurf_kernels.hpp"

namespace rfs {

// deconit's Gaussian: exp(-0.25 (2 pi f / f0)^2), f = k / (nft dt), pi = atan(1.0)*4. (float32), deconit.f90:24-29
__device__ __forceinline__ double rft_gauss(const RfFreq& f, int k) {
    double freq = k / (f.nft * f.dt);
    double x = 2 * RF_PI32 * freq / f.f0;
    return exp(-0.25 * (x * x));
}

// pulse spectrum G(k) exp(-i k/(nft dt) pi 2 tshift) (apply_gaussian + shift_data at the end of deconit)
__global__ void k_rft_pulse_spec(RfFreq f, cplx* __restrict__ spec)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= f.n2) return;
    double ph = k / (f.nft * f.dt) * RF_PI32 * 2 * f.t0;
    double s, c; sincos(ph, &s, &c);
    cplx v = rft_gauss(f, k) * C(c, -s);
    if (k == 0 || k == f.n2 - 1) v.im = 0.0;
    spec[k] = v;
}

// per chain: spectra of the three chain-level inverse transforms
//   [0] G^2 |R21'|^2            -> Aw of the forward trace      (w = irfft(R21))
//   [1] G^2 R22' conj(R21')     -> cuw0 of the forward trace    (u = irfft(R22))
//   [2] G^2 |(R21^2)'|^2        -> Aw of the partial traces     (w = irfft(R21^2))
// (' : imaginary parts of DC / Nyquist dropped, as the c2r that makes u, w real does), and
// S0f[chain] = sum_k c_k |uflt^_k|^2 of the forward trace (Parseval numerator).
__global__ void __launch_bounds__(256)
k_rft_chain_spectra(RfFreq f, const double* __restrict__ RR, cplx* __restrict__ spec3, double* __restrict__ S0f)
{
    __shared__ double red[4];
    int chain = blockIdx.x, tid = threadIdx.x;
    const double* rr = RR + (size_t)chain * 4 * f.n2p;
    cplx* o = spec3 + (size_t)chain * 3 * f.n2;
    double acc = 0.0;
    for (int k = tid; k < f.n2; k += blockDim.x) {
        cplx r21 = C(rr[k], rr[f.n2p + k]), r22 = C(rr[2 * f.n2p + k], rr[3 * f.n2p + k]);
        cplx sq = r21 * r21;
        const bool edge = (k == 0 || k == f.n2 - 1);
        if (edge) { r21.im = 0.0; r22.im = 0.0; sq.im = 0.0; }
        double g = rft_gauss(f, k), g2 = g * g;
        o[k] = C(g2 * norm2(r21));
        cplx cu = g2 * (r22 * conj(r21));
        if (edge) cu.im = 0.0;
        o[f.n2 + k] = cu;
        o[2 * f.n2 + k] = C(g2 * norm2(sq));
        acc += (edge ? 1.0 : 2.0) * g2 * norm2(r22);
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) s += red[i];
        S0f[chain] = s;                  // Parseval numerator: sum(uflt^2) = s / nft
    }
}

// Partial traces: spectrum of cuw0 for every (parameter class, layer),
//   G^2 num' conj((R21^2)'),  num = R22_m R21 - R21_m R22   (RFModule.f90:131-133),
// from ONE top-down sweep of one column (the two unit-seed columns for dR21 and dR22 combined up front), plus the
// Parseval sums S0 = sum(uflt^2) per trace (wave butterflies, one partial per wave like pass B).
// specp: [chain][4][n][n2] complex;  S0p: [chain][npart][4][n].
template <bool TAIL>
__global__ void __launch_bounds__(256)
k_rft_partial_spectra(int nchain, int n, RfFreq f, const RfLayer* __restrict__ lc, const double* __restrict__ RR,
                      const double* __restrict__ Rs, int npart, cplx* __restrict__ specp, double* __restrict__ S0p)
{
    int chain, k, part;
    bool live = true;
    if (TAIL) {
        chain = blockIdx.x * blockDim.x + threadIdx.x; k = f.n2 - 1; part = npart - 1;
        if (chain >= nchain) return;
    } else {
        chain = blockIdx.y; k = blockIdx.x * blockDim.x + threadIdx.x;
        part = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        if (k >= f.n2 - 1) { live = false; k = f.n2 - 2; }
    }
    const RfLayer* L = lc + (size_t)chain * n;
    const size_t n2p = f.n2p;
    const double* rr = RR + (size_t)chain * 4 * n2p + k;
    cplx r21 = C(rr[0], rr[n2p]), r22 = C(rr[2 * n2p], rr[3 * n2p]);
    cplx omega = C(rf_wk(f, k), -f.sigma), kk = f.p * omega;
    const bool edge = (k == 0 || k == f.n2 - 1);
    cplx sq = r21 * r21;
    if (edge) sq.im = 0.0;
    const double g = rft_gauss(f, k), g2 = g * g, ck = edge ? 1.0 : 2.0;
    const cplx Q = g2 * conj(sq);
    const size_t nkp = f.nkp;                                   // (= n2p: the time-domain method runs without the band limit)
    const double* rs = Rs + ((size_t)chain * (n - 1)) * 8 * nkp + k;
    const int c21 = (f.rf_type == 1) ? 0 : 1, c22 = 1 - c21;
    cplx* out = specp + (size_t)chain * 4 * n * f.n2 + k;
    // num is linear in the two unit-seed columns and both see the same layer matrices: carry their combination
    // (+-i R21) e_c22 - R22 e_c21 through ONE sweep
    V4 y;
    y.v[0] = y.v[1] = y.v[2] = y.v[3] = C(0.0);
    y.v[c22] = (f.rf_type == 1) ? mul_i(r21) : -mul_i(r21);
    y.v[c21] = -r22;
    double acc[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    const int lane = threadIdx.x & 63;
    double* sp = S0p + ((size_t)chain * npart + part) * 4 * n;
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
        double e4[4];
#pragma unroll
        for (int ip = 0; ip < 4; ip++) {
            cplx num = T[ip];
            if (num.re != num.re || num.im != num.im) num = C(0.0);  // NaN scrub, RFModule.f90:698-703
            if (edge) num.im = 0.0;
            cplx S = Q * num;
            if (edge) S.im = 0.0;
            double e = live ? ck * g2 * norm2(num) : 0.0;
            if (live) out[((size_t)ip * n + j) * f.n2] = S;
            e4[ip] = e;
            if (TAIL) sp[(size_t)ip * n + j] = e;
        }
        if (!TAIL) {
            double t4[4];
            wave_sum4_uniform(e4, t4);
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
                for (int ip = 0; ip < 4; ip++) sp[(size_t)ip * n + j] = acc[ip][s];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// The deconvolution loop: one wavefront per trace, WPB traces of ONE chain per block (the host guarantees
// traces_per_chain % WPB == 0), the chain's autocorrelation staged in LDS.
//   cuw0ts : c2r output of the trace's correlation spectrum (unnormalised; first half used)
//   awts   : c2r output of the chain's |wflt^|^2 spectrum (unnormalised)
//   S0parts: Parseval partial sums  sum_k c_k |uflt^_k|^2  (nS0 parts), so that sum(uflt^2) = sum / nft
// Lane l owns lags j = i*64 + l.  Outputs (each optional): Pout[trace][nft/2] spike train, gout[trace] =
// sum_spikes a * Cres[chain][lag], nit_out[trace] = iterations used.
// ---------------------------------------------------------------------------------------
// wave-wide max of a non-negative f64 on the VALU.  The bit pattern of a non-negative double orders like an unsigned
// integer, so the maximum is taken in two 32-bit passes (high words, then the low words of the lanes that hold the
// maximal high word): v_max_u32 takes a DPP operand directly, one instruction per butterfly step, where an f64 max
// needs two DPP moves, a canonicalisation and the max (no LDS crossbar traffic either way).  Result is wave-uniform.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned rft_dpp_umax(unsigned v) {
    unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);   // 0 = identity of umax
    return o > v ? o : v;
}
__device__ __forceinline__ unsigned rft_wave_umax(unsigned v) {
    v = rft_dpp_umax<0xb1, 0xf>(v);      // quad_perm [1,0,3,2]
    v = rft_dpp_umax<0x4e, 0xf>(v);      // quad_perm [2,3,0,1]
    v = rft_dpp_umax<0x114, 0xf>(v);     // row_shr:4
    v = rft_dpp_umax<0x118, 0xf>(v);     // row_shr:8
    v = rft_dpp_umax<0x142, 0xa>(v);     // row_bcast:15 -> rows 1, 3
    v = rft_dpp_umax<0x143, 0xc>(v);     // row_bcast:31 -> rows 2, 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ double rft_wave_max_uniform(double v) {      // v >= 0 (or NaN, which wins and ends the loop)
    const unsigned hi = (unsigned)__double2hiint(v), lo = (unsigned)__double2loint(v);
    const unsigned mh = rft_wave_umax(hi);
    const unsigned ml = rft_wave_umax(hi == mh ? lo : 0u);
    return __hiloint2double((int)mh, (int)ml);
}
__device__ __forceinline__ double rft_max_abs(double a, double b) {     // max(|a|, |b|) in one instruction
    double r;
    asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// FULL: nft >= 128, every lane owns exactly NPL lags (no guards, the NPL autocorrelation reads of an update are
// issued together: consecutive lags of a lane are 512 B apart in the doubled LDS copy, which is what
// ds_read2st64_b64 addresses).  WANT_P: the spike train is wanted (B1 kernels); the B2 gradient only needs gout.
template <int NPL, int WPB, bool FULL, bool WANT_P>
__global__ void __launch_bounds__(64 * WPB)
k_rft_deconv(int ntrace, int trace_per_chain, RfFreq f, const double* __restrict__ cuw0ts, size_t cuw_stride,
             const double* __restrict__ awts, size_t aw_stride, const double* __restrict__ S0parts, int nS0,
             size_t s0_chain_stride, size_t s0_part_stride, const double* __restrict__ Cres,
             double* __restrict__ Pout, double* __restrict__ gout, int* __restrict__ nit_out)
{
    extern __shared__ double aw2[];                               // [2 nft] autocorrelation of wflt (unnormalised), twice
    const int trace0 = blockIdx.x * WPB;
    const int trace = trace0 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int chain = trace0 / trace_per_chain;
    const int nft = f.nft, half = nft >> 1;
    double* cres_l = aw2 + 2 * nft;                               // [half] correlation of the residual with the pulse
    {
        const double* ag = awts + (size_t)chain * aw_stride;
        for (int i = threadIdx.x; i < nft; i += 64 * WPB) { double v = ag[i]; aw2[i] = v; aw2[i + nft] = v; }
        if (Cres) {
            const double* cg = Cres + (size_t)chain * half;
            for (int i = threadIdx.x; i < half; i += 64 * WPB) cres_l[i] = cg[i];
        }
    }
    __syncthreads();
    if (trace >= ntrace) return;
    const int tl = trace - chain * trace_per_chain;
    const double dt = f.dt, inft = 1.0 / nft;
    const double* cu = cuw0ts + (size_t)trace * cuw_stride;
    double S0 = 0.0;
    {
        const double* sp = S0parts + (size_t)chain * s0_chain_stride + tl;
        for (int p = 0; p < nS0; p++) S0 += sp[(size_t)p * s0_part_stride];
        S0 = S0 * inft;                                           // sum(uflt^2)
    }
    const double Aw0 = aw2[0] * inft;                             // sum(wflt^2)
    const double invpw = 1. / Aw0 / dt, invpu = 1. / S0 / dt;     // deconit.f90:162-163
    double cuw[NPL], P[WANT_P ? NPL : 1];
#pragma unroll
    for (int i = 0; i < NPL; i++) {
        int j = i * 64 + lane;
        cuw[i] = (FULL || j < half) ? cu[j] * inft * dt : 0.0;    // mycorrelate(...) * dt, :174-175
        if (WANT_P) P[i] = 0.0;
    }
    double S = S0, sumsq_i = 1.0, d_error = 100 * invpw + 0.001, gacc = 0.0;
    const double rA = 1.0 / aw2[0];
    const double ka = invpw / dt, ks = dt * invpu;                // per-iteration products of :177, :184 hoisted
    const double* awl = aw2 + nft + lane;                         // lag (j - bj) mod nft of lane's lag i: awl[64 i - bj]
    int it = 0;
    for (; it < 200; it++) {
        if (fabs(d_error) <= 0.001) break;                        // :172
        // maxloc(abs(cuw(1:nft/2))): the maximum (exact: max does not round), then its first position
        double bv;
        if (NPL == 1) bv = fabs(cuw[0]);                          // idle lanes of a short trace hold 0
        else {
            bv = rft_max_abs(cuw[0], cuw[1]);
#pragma unroll
            for (int i = 2; i < NPL; i++) bv = rft_max_abs(bv, cuw[i]);
        }
        bv = rft_wave_max_uniform(bv);
        if (!(bv > 0.0)) { it++; break; }                         // nothing left to fit: P can no longer change
        int bj = -1; double c = 0.0;
#pragma unroll
        for (int i = 0; i < NPL; i++) {
            if (bj < 0) {
                unsigned long long hit = __builtin_amdgcn_ballot_w64((FULL || i * 64 + lane < half) && fabs(cuw[i]) == bv);
                if (hit) {
                    int l = __ffsll((long long)hit) - 1;
                    bj = i * 64 + l;
                    c = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(cuw[i]), l),
                                         __builtin_amdgcn_readlane(__double2loint(cuw[i]), l));
                }
            }
        }
        const double cr = Cres ? cres_l[bj] : 0.0;                // fetched early, used after the update
        const double a = c * ka;                                  // cuw(idx) * invpw / dt, :177
        const double r = c * rA;
        const double* ar = awl - bj;
#pragma unroll
        for (int i = 0; i < NPL; i++)
            if (FULL || i * 64 + lane < half) cuw[i] -= r * ar[64 * i];
        if (WANT_P) {
            const double ap = (lane == (bj & 63)) ? a : 0.0;      // bj is wave-uniform: one scalar-selected add
            const int bi = bj >> 6;
#pragma unroll
            for (int i = 0; i < NPL; i++)
                if (bi == i) P[i] += ap;
        }
        gacc += a * cr;
        S -= a * c;
        double sumsq = S * ks;                                    // sum(rflt**2) * dt * invpu, :184
        d_error = 100. * (sumsq_i - sumsq);
        sumsq_i = sumsq;
    }
    if (WANT_P) {
#pragma unroll
        for (int i = 0; i < NPL; i++) {
            int j = i * 64 + lane;
            if (FULL || j < half) Pout[(size_t)trace * half + j] = P[i];
        }
    }
    if (lane == 0) {
        if (gout) gout[trace] = gacc;
        if (nit_out) nit_out[trace] = it;
    }
}

// Traces longer than 4096 samples (nft >= 8192): the lags of a trace no longer fit one wavefront's registers nor the
// autocorrelation the LDS, so ONE BLOCK owns the trace -- thread t the lags t, t + blockDim, ... -- with cuw updated in
// place in the c2r output it arrives in (scratch), the autocorrelation read through L2 and the arg-max combined
// across the block's waves in LDS.  Same arithmetic, statement for statement, as k_rft_deconv; any power-of-two nft.
template <bool WANT_P>
__global__ void __launch_bounds__(1024)
k_rft_deconv_big(int ntrace, int trace_per_chain, RfFreq f, double* __restrict__ cuw0ts, size_t cuw_stride,
                 const double* __restrict__ awts, size_t aw_stride, const double* __restrict__ S0parts, int nS0,
                 size_t s0_chain_stride, size_t s0_part_stride, const double* __restrict__ Cres,
                 double* __restrict__ Pout, double* __restrict__ gout)
{
    __shared__ unsigned long long wkey[2][16];
    __shared__ int widx[2][16];
    __shared__ double wval[2][16];
    const int trace = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
    const int chain = trace / trace_per_chain, tl = trace - chain * trace_per_chain;
    const int nft = f.nft, half = nft >> 1, mask = nft - 1;
    const double dt = f.dt, inft = 1.0 / nft;
    double* cu = cuw0ts + (size_t)trace * cuw_stride;
    const double* aw = awts + (size_t)chain * aw_stride;
    double* P = WANT_P ? Pout + (size_t)trace * half : nullptr;
    double S0 = 0.0;
    {
        const double* sp = S0parts + (size_t)chain * s0_chain_stride + tl;
        for (int p = 0; p < nS0; p++) S0 += sp[(size_t)p * s0_part_stride];
        S0 = S0 * inft;
    }
    const double Aw0 = aw[0] * inft;
    const double invpw = 1. / Aw0 / dt, invpu = 1. / S0 / dt;
    for (int j = tid; j < half; j += blockDim.x) {
        cu[j] = cu[j] * inft * dt;
        if (WANT_P) P[j] = 0.0;
    }
    double S = S0, sumsq_i = 1.0, d_error = 100 * invpw + 0.001, gacc = 0.0;
    const double rA = 1.0 / aw[0];
    const double ka = invpw / dt, ks = dt * invpu;
    for (int it = 0; it < 200; it++) {
        if (fabs(d_error) <= 0.001) break;                        // block-uniform: every thread carries the same scalars
        // first position of the largest |cuw|; |v| compared through its bit pattern (a NaN wins and ends the loop)
        unsigned long long key = 0ull; int idx = -1; double val = 0.0;
        for (int j = tid; j < half; j += blockDim.x) {
            double v = cu[j];
            unsigned long long k = (unsigned long long)__double_as_longlong(fabs(v));
            if (k > key) { key = k; idx = j; val = v; }
        }
        for (int off = 32; off >= 1; off >>= 1) {
            unsigned long long ok = __shfl_xor(key, off); int oi = __shfl_xor(idx, off); double ov = __shfl_xor(val, off);
            if (ok > key || (ok == key && oi >= 0 && (idx < 0 || oi < idx))) { key = ok; idx = oi; val = ov; }
        }
        const int b = it & 1;
        if (lane == 0) { wkey[b][wv] = key; widx[b][wv] = idx; wval[b][wv] = val; }
        __syncthreads();
        key = wkey[b][0]; int bj = widx[b][0]; double c = wval[b][0];
        for (int w = 1; w < nw; w++) {
            unsigned long long ok = wkey[b][w]; int oi = widx[b][w];
            if (ok > key || (ok == key && oi >= 0 && (bj < 0 || oi < bj))) { key = ok; bj = oi; c = wval[b][w]; }
        }
        const double bv = __longlong_as_double((long long)key);
        if (!(bv > 0.0)) break;
        const double cr = Cres ? Cres[(size_t)chain * half + bj] : 0.0;
        const double a = c * ka;
        const double r = c * rA;
        for (int j = tid; j < half; j += blockDim.x) cu[j] -= r * aw[(j - bj) & mask];
        if (WANT_P && tid == 0) P[bj] += a;
        gacc += a * cr;
        S -= a * c;
        double sumsq = S * ks;
        d_error = 100. * (sumsq_i - sumsq);
        sumsq_i = sumsq;
    }
    if (tid == 0 && gout) gout[trace] = gacc;
}

// k_rft_synth for nft >= 8192: the pulse read through L2, the (at most 200) spikes compacted into LDS in lag order
__global__ void __launch_bounds__(256)
k_rft_synth_big(int ntrace, RfFreq f, const double* __restrict__ P, const double* __restrict__ gshts,
                double* __restrict__ out, size_t ostride)
{
    __shared__ double amp[256];
    __shared__ int lag[256];
    __shared__ int nsp;
    const int nft = f.nft, half = nft >> 1, mask = nft - 1;
    const int trace = blockIdx.x, tid = threadIdx.x;
    if (tid < 64) {
        int m = 0;
        const double* p = P + (size_t)trace * half;
        for (int base = 0; base < half; base += 64) {
            double v = p[base + tid];
            unsigned long long bal = __ballot(v != 0.0);
            int pos = m + __popcll(bal & ((1ull << tid) - 1ull));
            if (v != 0.0 && pos < 256) { amp[pos] = v; lag[pos] = base + tid; }
            m += __popcll(bal);
        }
        if (tid == 0) nsp = m < 256 ? m : 256;
    }
    __syncthreads();
    const int m = nsp;
    for (int t = tid; t < f.nt; t += blockDim.x) {
        double s = 0.0;
        for (int q = 0; q < m; q++) s += amp[q] * (gshts[(t - lag[q]) & mask] / nft);
        out[(size_t)trace * ostride + t] = s;
    }
}

// k_rft_resid_cres for nft >= 8192: the residual staged through LDS in tiles of 2048 samples, Cres accumulated in place
// (the running sum of a lag continues from the stored value, so the summation order over t is unchanged)
__global__ void __launch_bounds__(256)
k_rft_resid_cres_big(RfFreq f, const double* __restrict__ dsyn, int ndata, const double* __restrict__ dobs,
                     const double* __restrict__ gshts, double* __restrict__ misfit_rf, double* __restrict__ Cres)
{
    __shared__ double res[2048];
    __shared__ double red[4];
    const int nft = f.nft, half = nft >> 1, mask = nft - 1;
    const int chain = blockIdx.x, tid = threadIdx.x;
    double acc = 0.0;
    for (int t0 = 0; t0 < f.nt; t0 += 2048) {
        const int m = min(2048, f.nt - t0);
        __syncthreads();
        for (int t = tid; t < m; t += blockDim.x) {
            double r = dsyn[(size_t)chain * ndata + t0 + t] - dobs[t0 + t];
            res[t] = r; acc += r * r;
        }
        __syncthreads();
        for (int i = tid; i < half; i += blockDim.x) {
            double s = t0 ? Cres[(size_t)chain * half + i] : 0.0;
            for (int t = 0; t < m; t++) s += (gshts[(t0 + t - i) & mask] / nft) * res[t];
            Cres[(size_t)chain * half + i] = s;
        }
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) s += red[i];
        misfit_rf[chain] = 0.5 * s;
    }
}

// out[trace][t] = sum_j P[j] gsh[(t - j) mod nft], t < nt  (apply_gaussian + shift_data of the spike train)
// gshts: c2r output of the pulse spectrum (unnormalised).  out row stride = ostride.
__global__ void __launch_bounds__(256)
k_rft_synth(int ntrace, RfFreq f, const double* __restrict__ P, const double* __restrict__ gshts,
            double* __restrict__ out, size_t ostride)
{
    extern __shared__ double sh[];            // [nft] pulse, [half] amplitudes, [half] lags (as int)
    const int nft = f.nft, half = nft >> 1, mask = nft - 1;
    double* pulse = sh; double* amp = sh + nft; int* lag = (int*)(amp + half);
    __shared__ int nsp;
    const int trace = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < nft; i += blockDim.x) pulse[i] = gshts[i] / nft;
    if (tid < 64) {                                   // wave 0 compacts the non-zero spikes, in lag order
        int m = 0;
        const double* p = P + (size_t)trace * half;
        for (int base = 0; base < half; base += 64) {
            int j = base + tid;
            double v = (j < half) ? p[j] : 0.0;
            unsigned long long bal = __ballot(v != 0.0);
            int pos = m + __popcll(bal & ((1ull << tid) - 1ull));
            if (v != 0.0) { amp[pos] = v; lag[pos] = j; }
            m += __popcll(bal);
        }
        if (tid == 0) nsp = m;
    }
    __syncthreads();
    const int m = nsp;
    for (int t = tid; t < f.nt; t += blockDim.x) {
        double s = 0.0;
        for (int q = 0; q < m; q++) s += amp[q] * pulse[(t - lag[q]) & mask];
        out[(size_t)trace * ostride + t] = s;
    }
}

// per chain: residual r = rf - dobs, misfit_rf = 0.5 sum r^2 (model_rf.py:168-196) and
// Cres[i] = sum_{t<nt} gsh[(t - i) mod nft] r[t], i < nft/2.
__global__ void __launch_bounds__(256)
k_rft_resid_cres(RfFreq f, const double* __restrict__ dsyn, int ndata, const double* __restrict__ dobs,
                 const double* __restrict__ gshts, double* __restrict__ misfit_rf, double* __restrict__ Cres)
{
    extern __shared__ double sh[];            // [nft] pulse, [nt] residual
    __shared__ double red[4];
    const int nft = f.nft, half = nft >> 1, mask = nft - 1;
    double* pulse = sh; double* res = sh + nft;
    const int chain = blockIdx.x, tid = threadIdx.x;
    double acc = 0.0;
    for (int i = tid; i < nft; i += blockDim.x) pulse[i] = gshts[i] / nft;
    for (int t = tid; t < f.nt; t += blockDim.x) {
        double r = dsyn[(size_t)chain * ndata + t] - dobs[t];
        res[t] = r; acc += r * r;
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) s += red[i];
        misfit_rf[chain] = 0.5 * s;
    }
    for (int i = tid; i < half; i += blockDim.x) {
        double s = 0.0;
        for (int t = 0; t < f.nt; t++) s += pulse[(t - i) & mask] * res[t];
        Cres[(size_t)chain * half + i] = s;
    }
}

}  // namespace rfs
