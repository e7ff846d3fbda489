// librfsurf_hip.so -- C ABI (include/rfsurf.h) over the gfx950 kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -fPIC -shared rfsurf_hip.hip -lrocfft  (see build.py)
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/rfsurf.h"
#include "rfsurf_kernels.hpp"
#include "rf_time_kernels.hpp"

using namespace rfs;

#ifndef RFS_F32_DEFAULT
#define RFS_F32_DEFAULT 1      // (A/B builds: -DRFS_F32_DEFAULT=0)
#endif
namespace {

constexpr int COLD_AUTO_CHAINS = 64, COLD_MAX_CHAINS = 512;   // batches whose declined chains take the search without a prediction (k_swd_cold_scan): by default / with "swd_cold_scan" 1
constexpr int EXACT_COOP_MAX = 8192;   // (group, chain) pairs of the reference-root stage up to which a group takes 16 lanes (k_swd_exact_coop): ~1 500 wavefronts hold 6 000 groups at once
constexpr int RFS_BG_SLOTS = 8;      // sets of hand-back flags / lists (and events): background searches of that many steps may be in flight

struct Buf {
    void* p = nullptr;
    size_t cap = 0;
    template <class T> T* as() const { return (T*)p; }
};

struct FftPlan { rocfft_plan plan = nullptr; rocfft_execution_info info = nullptr; void* work = nullptr; };

}  // namespace

// Schedule of a step, MEASURED once per shape: the candidates -- everything on shared CUs, or the CU partition with
// 0 / some / more of the first periods' eigenfunction kernels moved onto the RF half -- are each run (as ordinary
// evaluations: every schedule returns bit-identical results) with one HIP-event pair around the whole step, twice; the
// fastest one is kept for the shape.  The first call of a shape (buffer and FFT-plan allocation) is not timed.
struct StepCand { bool part; int early; float ms; };
struct StepCalib {
    int stage = -1;                      // -1 never seen; 0 .. 2 * ncand: timed calls launched so far; CALIB_DONE: decided
    int harvested = 0;                   // timed calls whose event pair has been read (never waited for: hipEventQuery)
    std::vector<hipEvent_t> ev;          // [2 * ncand][begin, end]
    std::vector<StepCand> cand;
    bool part = true;
    int early_items = 0;
};
constexpr int CALIB_DONE = 1 << 20;

struct rfs_ctx {
    int device = 0, max_chains = 0, max_layers = 0;
    std::map<std::tuple<int, int, int, int, int>, StepCalib> calib;   // key: chains, layers, nft, periods, sequences
    size_t rf_scratch_budget = (size_t)4 << 30;   // bytes of pass-A row scratch (Rs) per chain tile of the fused gradient
    // user/main stream; SWD search stream; CU-partitioned pair (search on one half of the chip, RF on the other)
    hipStream_t stream = nullptr, stream2 = nullptr, stream2m = nullptr, stream3 = nullptr;
    hipStream_t stream_l = nullptr;     // Love root search beside the Rayleigh one (unpartitioned steps)
    hipStream_t stream_w = nullptr;     // the grid walk of irregular sequences beside the reference-root stage (background form of the flow entries)
    hipEvent_t ev_wk[2] = {nullptr, nullptr};
    hipEvent_t ev_lf = nullptr, ev_lj = nullptr;
    bool own_stream = false;
    int early_eigen = -1;      // periods whose eigenfunction kernels run early on the RF half: -1 automatic, 0 off
    int cu_split = 1;          // 0: never partition; 1/2: partition (contiguous / even-odd mask bits) when the
                               // cooperative search fits on half of the CUs
    int ncu = 0;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join3 = nullptr;
    std::string err;
    // joint configuration
    bool configured = false;
    int n = 0, mode = 0, ndata = 0;
    int ntw[4] = {0, 0, 0, 0};   // rows of the Rc, Rg, Lc, Lg blocks
    bool rg_alias = false;       // tRg == tRc: the Rg block's pass at T reads the Rc block's items (make_plan)
    bool lg_alias = false;       // tLg == tLc likewise
    int sphere = 0;
    bool has_rf = false, has_swd = false;
    RfFreq f{};
    double wt = 1.0;
    Buf d_tw[4], d_dobs;
    // workspaces
    int swd_lanes = 0;     // lanes per chain in the root search (0 = pick from nchain / nlayer)
    int share_rc_rg = 1;   // option "share_rc_rg": 0 = never alias the Rg block's central pass to the Rc block (tests)
    int swd_speculate = -1; // wavefronts per block that look ahead in the scan of the lanes-per-chain search (-1 = automatic, 1 / 2 / 4)
    int swd_segments = -1; // segments of the vector recurrence in the lanes-per-chain search (-1 = automatic, 1 / 2 / 4)
    Buf d_minv; bool has_minv = false;                              // diagonal inverse mass of the leapfrog kernels
    // warm start of the root search inside trajectories (k_swd_warm): roots / kernels / model of the previous evaluation
    int rf_peel = -1;          // option "rf_row_peeling": pass B peels the layers off pass A's final row instead of reading stored rows: -1 / 1 = for the chains whose growth exponent allows it (decided on the device), 0 = never, 2 = always (diagnostics)
    int rf_band_digits = 13;   // option "rf_band_limit_digits": adjoint band limit at 1e-digits * water (0 = off)
    int rf_band_floor = 8;     // option "rf_band_floor_digits": the limit may move down to a multiple of 64 bins, never below this
    int warm_opt = 1;          // option "swd_warm_start": 0 off, 1 trajectory entries, 2 also the plugin entries
    int warm_serial = 0;       // option "swd_warm_serial": warm-started steps on ONE stream (1) or SWD beside RF (0)
    int exact_final = 0;       // option "swd_exact_final": first and last evaluation of a trajectory by the full search
    int warm_exact = 1;        // option "swd_warm_exact": behind the warm start, the reference's own refinement inside the reference's scan cell (k_swd_exact): the reference's roots
    int flow_async = 0;        // option "flow_async_handback": in the flow entries a chain handed back to the full search sits out the step (its search runs beside the next one) instead of holding every chain
    unsigned wpar = 0;         // slot of the hand-back lists / flags in use (RFS_BG_SLOTS sets: the searches of earlier steps still read theirs)
    bool bg_busy[RFS_BG_SLOTS] = {};   // a background search recorded on the slot's event has not been seen complete yet
    unsigned bg_ready = 0;     // slots whose background search was complete when the evaluation being launched started (bit mask)
    bool flow_cur = false;     // the evaluation being launched is a flow step (rfs_flow_step2): `fpend` applies, and with flow_async its handed-back chains stay in the background
    bool last_async = false;   // ... and did (a warm-started step with a side stream)
    int fpend_nchain = 0;
    const double* flow_x = nullptr;   // the state a flow step last advanced (its x array): what the warm start and `fpend` describe
    bool warm_coop = true;       // option "swd_warm_last_round_coop": 16 lanes per search in the last round of k_swd_warm (k_swd_warm_coop)
    int warm_budgets = 303;      // option "swd_warm_round_budgets": evaluations a lane may spend in round 1 / 2 / 3 of k_swd_warm (b1 + 100 b2 + 10000 b3; the last round has no limit); 0: one round
    bool warm_feedback = true;   // option "swd_warm_feedback": last step's prediction error corrects this step's prediction (SwdWarm::ferr)
    Buf wferr;
    bool warm_widen = true;    // option "swd_warm_widen": the warm search may bracket beyond its trust radius (the grid walk then vouches)
    bool flow_skip_idle = true;   // option "flow_skip_idle": idle chains of a flow step are neither continued nor handed back
    const int *f_rem = nullptr, *f_fresh = nullptr, *f_ok = nullptr;   // the flow state's arrays during a flow step (k_swd_warm: idle chains)
    Buf frec;                  // the device word of rfs_flow_records: records handed out so far (k_flow_post; a ring, never reset between steps)
    Buf fpend;                 // [chain] 1: handed back in the previous flow step (k_flow_post) -- no drift, no warm start this time: its roots are the background search's
    hipEvent_t ev_bg[RFS_BG_SLOTS] = {};
    // TWO run-up periods, origins accepted to 1e-7 c.  Round 5 measured ONE period with 5e-7 c: a sixth less work in the stage
    // (-3.7 % per step), and over 3 072 + 3 072 burned-in bench chains against the oracle (phase velocities only,
    // scripts/flow_parity_stats.py) the same parity figures -- but on random configurations with group velocities (which
    // difference roots at neighbouring periods: a root one float32 step off counts fifty-fold) 79 instead of 1 of 518 473 roots
    // differ from the sequential search's and the misfit is off by up to 5.1e-5 instead of 6.0e-6 (scripts/warm_fuzz_soak.py
    // 8100..8399).  Parity first: the one-period setting is an option ("swd_exact_runup" 1 + "swd_exact_origin_tol_e9" 500).
    float exact_origin_tol = 1.0e-7f;   // option "swd_exact_origin_tol_e9" (EXACT_ORIGIN_TOL; 5e-7 goes with ONE run-up period)
    int cold_again = 1;             // "swd_cold_again": a chain handed back in the evaluation before goes straight to the search without a prediction
    int need_prev_par = -1, need_prev_nchain = 0;
    Buf wcold;
    int cold_first = 8;             // "swd_cold_first": batches of up to that many chains do not try the warm search at all
    int cold_scan = -1;             // "swd_cold_scan": -1 = for foreground hand-backs of up to COLD_AUTO_CHAINS chains, 0 off, 1 up to COLD_MAX_CHAINS
    Buf cold_roots, cold_nroot, cold_s0, cold_ticket;
    bool counters_zeroed = false;   // k_prep_joint of the evaluation being launched cleared the list counters (wspc, xspc)
    int walk_window = 2;       // option "swd_walk_window": periods around an anomalous one that walk the reference's grid (-1: the whole sequence)
    int exact_budget = 44;     // option "swd_exact_budget": evaluations a lane of k_swd_exact may spend before its group goes on to the 16-lane launch (0: one round)
    Buf xsp, xspc;             // ... the saved machines of those groups (ExactSpill) and their counts (Rayleigh, Love)
    int exact_redo_runup = -1; // option "swd_exact_redo_runup": > "swd_exact_runup": a group whose run-up did not contract is done again with this many run-up periods (k_swd_exact_coop over a list) instead of handing its chain back; 0 = hand back
    Buf xredo;                 // ... the lists of those groups
    int exact_overlap = 0;     // option "swd_exact_overlap": the stage's second launch on the walk stream beside the eigenfunction pass of all items, its groups' eigenfunctions again afterwards (k_swd_eigen_groups; flow entries, background form).  Measured, round 6: 4.55 -> 4.67 ms -- the launch starves beside the pass and the RF sweeps, and the chain ends later than with both in a row: off
    hipEvent_t ev_x1 = nullptr;
    struct XGroups { const unsigned long long* item; const int* count; int cap, G; bool on; };
    XGroups xg[2] = {{nullptr, nullptr, 0, 0, false}, {nullptr, nullptr, 0, 0, false}};   // groups whose eigenfunctions follow the second launch (Rayleigh, Love)
    int exact_coop = 1;        // option "swd_exact_coop": 0 never, 1 (default) 16 lanes per group for small batches (k_swd_exact_coop), 2 always (tests)
    int exact_group_small = 2; // "swd_exact_group_small": periods per group in the small batches of "swd_cold_scan" (16 lanes per group)
    int exact_group = 4, exact_runup = 2;   // options "swd_exact_group" / "swd_exact_runup": periods per lane of k_swd_exact, run-up periods in front of them
    bool krn_ruled = false;    // the eigenfunction pass of the evaluation being launched stores chain-ruled kernels (joint_eval; B1 keeps the raw classes)
    bool warm_primed = false;  // croot / krn / xw describe the previous evaluation of the same nchain chains
    int warm_nchain = 0;
    int warm_est = 0;          // chains the last steps handed back to the full search (sizes the next fallback launch)
    int* h_wcount = nullptr;   // pinned mirror of the device-side count, copied back asynchronously (never waited for)
    Buf xw, dxT, crT, wvalid, wneed, wlist, wforce, wstats, wsgn, crs, craw, wilist, wlist2, wlist3, cwarm, fstat, wslope, wbetmx, wsg1, wspA, wspB, wspc;
    hipEvent_t ev_w[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // hand-overs between the SWD stream and its side stream (warm start)
    int swd_mode = 0, swd_mode_cur = 0;   // libsurf's `mode` of the joint configuration / of the evaluation being launched
    bool swd_water_cur = false;           // the batch being launched holds models with a water layer on top (B1 entries)
    int warm_nitems = 0;       // (sequence, period) items of the joint configuration's evaluation
    Buf spec3, ts3, S0f, S0p, pulse_spec, pulse_ts, Pbuf, Cres;   // time-domain RF (rf_time_kernels.hpp)
    double pulse_key[4] = {0, 0, 0, 0};
    Buf mdlc, mdlSR, mdlL, sphR, sphL, mdlcL;   // per-family search models / bldsph arrays (sphere, Love)
    Buf slist, scount;                    // chains of a peeling evaluation that kept stored rows (k_rf_passA -> k_rf_passB<., false>)
    int* h_scount = nullptr; int* d_hscount = nullptr; int stored_est = -1; unsigned speel_eval = 0;   // their number in an earlier evaluation (host-mapped word the device writes; -1 = unknown); evaluation parity of the two counters
    Buf RT, rstat;                        // final rows of pass A (row peeling, k_rf_passB<., true>); closure residual of the peeling
    int walk_dense = 1;                   // option "swd_walk_dense": the later periods' grid walk packed densely (k_swd_warm_walk_dense) instead of 8 speculative lanes per item
    int rf_mid_fused = 1;                 // option "rf_mid_fused": the middle section of the frequency-domain gradient in one kernel (k_rf_mid_fused) instead of k_rf_mid1 / rocFFT / k_rf_mid2 / rocFFT
    Buf twid; int twid_nft = 0;           // exp(-2 pi i j / nft), j <= nft / 2
    Buf gtab, etab; double mid_tab_key[6] = {0, 0, 0, 0, 0, 0};   // chain-independent factors of the fused middle section (k_rf_mid_tables)
    int rf_f32 = RFS_F32_DEFAULT;                       // option "rf_f32_beyond_band": pass A sweeps the frequencies beyond the gradient's band in float32 where that is provably enough
    Buf hi32, stat32;                     // [chain] pass A's choice; [66] chains swept again in f64 by k_rf_mid1, chains swept in float32 (64 slots)
    int rf_store_hyp = 0;                 // (measured, round 6: 4.73-4.81 -> 4.90-4.97 ms per step -- 110 instructions per layer of pass B saved, 2.9 GB of HBM traffic per step and their latency added: off) option "rf_store_hyp": pass A leaves exp / cos / sin of every (layer, band frequency) for pass B (row peeling only)
    Buf Hs; double* Hs_cur = nullptr;     // [chain][layer][6][nkp]; what the pass A of the evaluation being launched wrote (nullptr: nothing)
    int rf_peel_check = 0;                // option "rf_peel_check": pass B records the closure residual (statistic rf_peel_residual)
    Buf x, misfit, grad, dsyn, flag, lc, cr, mdl, RR, Rs, spec, tser, wres, W, wmax2, PG, mrf, croot, sflag, edone,
        cds, krn, ugr, b1a, b1b, b1c, b1d, b1e, b1f, b1g, specp, tserp, klbuf, bt;
    // leapfrog state
    Buf lx, lp, lU, lgrad, ldsyn, lflag;
    std::map<std::tuple<int, size_t, int>, FftPlan> plans;
    // timing
    // timing: every bracketed launch group gets its own event pair, recorded on the stream the
    // kernels run on; nothing synchronises until rfs_kernel_ms_sum() is called
    bool timing = false;
    unsigned timing_mask = ~0u;     // groups that get event pairs while timing is on (bit = rfs_kernel_id)
    std::vector<hipEvent_t> tev[RFS_K_COUNT];
    size_t tused[RFS_K_COUNT] = {};
    std::vector<hipEvent_t> tref[RFS_K_COUNT];   // per event pair: the start event of the flow step it belongs to (rfs_kernel_timeline), or nullptr
    hipEvent_t cur_step_ev = nullptr;
};

namespace {

bool g_rocfft_ready = false;

int fail(rfs_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}
#define HIPCHK(c, call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(c, RFS_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));     \
    } while (0)
#define FFTCHK(c, call)                                                                         \
    do {                                                                                        \
        rocfft_status s_ = (call);                                                              \
        if (s_ != rocfft_status_success)                                                        \
            return fail(c, RFS_ERR_HIP, std::string(#call) + ": rocfft status " + std::to_string((int)s_)); \
    } while (0)

int ensure(rfs_ctx* c, Buf& b, size_t bytes) {
    if (bytes <= b.cap) return RFS_OK;
    if (b.p) HIPCHK(c, hipFree(b.p));
    b.p = nullptr; b.cap = 0;
    HIPCHK(c, hipMalloc(&b.p, bytes));
    b.cap = bytes;
    return RFS_OK;
}
#define ENSURE(c, b, bytes) do { int r_ = ensure(c, b, bytes); if (r_) return r_; } while (0)
#define TRY(expr) do { int r_ = (expr); if (r_) return r_; } while (0)

void drop_plans(rfs_ctx* c) {
    for (auto& kv : c->plans) {
        if (kv.second.plan) rocfft_plan_destroy(kv.second.plan);
        if (kv.second.info) rocfft_execution_info_destroy(kv.second.info);
        if (kv.second.work) hipFree(kv.second.work);
    }
    c->plans.clear();
}

int get_plan(rfs_ctx* c, int nft, size_t batch, int inverse, FftPlan** out) {
    auto key = std::make_tuple(nft, batch, inverse);
    auto it = c->plans.find(key);
    if (it == c->plans.end()) {
        if (c->plans.size() >= 64) {       // callers quantise their batch sizes; this only bounds pathological use
            HIPCHK(c, hipStreamSynchronize(c->stream));
            drop_plans(c);
        }
        if (!g_rocfft_ready) { FFTCHK(c, rocfft_setup()); g_rocfft_ready = true; }
        FftPlan P;
        rocfft_plan_description d = nullptr;
        FFTCHK(c, rocfft_plan_description_create(&d));
        size_t n2 = (size_t)nft / 2 + 1, one = 1;
        if (inverse)
            FFTCHK(c, rocfft_plan_description_set_data_layout(d, rocfft_array_type_hermitian_interleaved,
                                                              rocfft_array_type_real, nullptr, nullptr, 1, &one, n2, 1, &one, (size_t)nft));
        else
            FFTCHK(c, rocfft_plan_description_set_data_layout(d, rocfft_array_type_real,
                                                              rocfft_array_type_hermitian_interleaved, nullptr, nullptr, 1, &one, (size_t)nft, 1, &one, n2));
        size_t len = (size_t)nft;
        FFTCHK(c, rocfft_plan_create(&P.plan, rocfft_placement_notinplace,
                                     inverse ? rocfft_transform_type_real_inverse : rocfft_transform_type_real_forward,
                                     rocfft_precision_double, 1, &len, batch, d));
        rocfft_plan_description_destroy(d);
        FFTCHK(c, rocfft_execution_info_create(&P.info));
        size_t wb = 0;
        FFTCHK(c, rocfft_plan_get_work_buffer_size(P.plan, &wb));
        if (wb) {
            HIPCHK(c, hipMalloc(&P.work, wb));
            FFTCHK(c, rocfft_execution_info_set_work_buffer(P.info, P.work, wb));
        }
        it = c->plans.emplace(key, P).first;
    }
    FFTCHK(c, rocfft_execution_info_set_stream(it->second.info, c->stream));
    *out = &it->second;
    return RFS_OK;
}

int run_fft(rfs_ctx* c, int nft, size_t batch, int inverse, void* in, void* out) {
    FftPlan* P = nullptr;
    TRY(get_plan(c, nft, batch, inverse, &P));
    void* ib[1] = {in};
    void* ob[1] = {out};
    FFTCHK(c, rocfft_execute(P->plan, ib, ob, P->info));
    return RFS_OK;
}

struct KTimer {   // brackets a group of launches with HIP events on the stream they run on
    rfs_ctx* c; int id; hipStream_t s; hipEvent_t e1 = nullptr;
    KTimer(rfs_ctx* c_, int id_, hipStream_t s_) : c(c_), id(id_), s(s_) {
        if (!c->timing || id < 0 || !((c->timing_mask >> id) & 1u)) return;
        auto& pool = c->tev[id];
        size_t& u = c->tused[id];
        while (pool.size() < u + 2) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; pool.push_back(e); }
        hipEventRecord(pool[u], s); e1 = pool[u + 1];
        if (id == RFS_K_FLOW_STEP) c->cur_step_ev = pool[u];
        auto& ref = c->tref[id];
        if (ref.size() < u / 2 + 1) ref.resize(u / 2 + 1);
        ref[u / 2] = c->cur_step_ev;
        u += 2;
    }
    ~KTimer() { if (e1) hipEventRecord(e1, s); }
};

int make_partition_streams(rfs_ctx* c) {
    if (c->stream2m) { hipStreamDestroy(c->stream2m); c->stream2m = nullptr; }
    if (c->stream3) { hipStreamDestroy(c->stream3); c->stream3 = nullptr; }
    if (c->cu_split == 0) return RFS_OK;
    int ncu = c->ncu, nw = (ncu + 31) / 32;
    std::vector<uint32_t> ma(nw, 0), mb(nw, 0);
    for (int i = 0; i < ncu; i++) {
        bool a = (c->cu_split == 1) ? (i < ncu / 2) : ((i & 1) == 0);
        (a ? ma : mb)[i / 32] |= 1u << (i % 32);
    }
    HIPCHK(c, hipExtStreamCreateWithCUMask(&c->stream2m, nw, ma.data()));
    HIPCHK(c, hipExtStreamCreateWithCUMask(&c->stream3, nw, mb.data()));
    return RFS_OK;
}

int check_rf(rfs_ctx* c, const rfs_rf_params* p) {
    if (!p) return fail(c, RFS_ERR_ARG, "rf params missing");
    if (p->method != RFS_RF_TIME && p->method != RFS_RF_FREQ && p->method != RFS_RF_TIME_PAR) return fail(c, RFS_ERR_ARG, "bad rf method");
    if (p->rf_type != RFS_RF_P && p->rf_type != RFS_RF_S) return fail(c, RFS_ERR_ARG, "rf_type should be one of [P,p,S,s]");
    if (p->nt < 2 || p->dt <= 0 || p->ray_p <= 0) return fail(c, RFS_ERR_ARG, "bad rf scalars");
    return RFS_OK;
}

RfFreq make_freq(const rfs_rf_params& p, int fwd_order) {
    RfFreq f{};
    f.nft = rf_nextpow2(p.nt); f.n2 = f.nft / 2 + 1; f.n2p = (f.n2 + 15) / 16 * 16;
    f.nk = f.n2; f.nkp = f.n2p;
    f.nt = p.nt; f.dt = p.dt; f.p = p.ray_p; f.f0 = p.gauss; f.water = p.water;
    f.rf_type = p.rf_type;
    f.t0 = (p.rf_type == RFS_RF_S) ? -p.time_shift : p.time_shift;     // src/RF/main.cpp:35
    f.sigma = 1.0 / p.dt / f.nft * 4.;                                  // RFModule.f90:381
    f.fwd_order = fwd_order;
    f.method = p.method; f.pi64 = 0;
    if (p.method != RFS_RF_FREQ) {       // real frequency axis (cal_rf_time :173, cal_rf_par_time_all :112)
        f.sigma = 0.0;
        f.pi64 = (!fwd_order && p.method == RFS_RF_TIME) ? 1 : 0;
    }
    return f;
}

int rf_block(const RfFreq& f) {
    int m = f.n2 - 1;
    if (m >= 256) return 256;
    return m < 64 ? 64 : (m / 64) * 64;
}
int rf_chunks(const RfFreq& f) { int bs = rf_block(f); return (f.n2 - 1 + bs - 1) / bs; }
int rf_nparts(const RfFreq& f) { return rf_chunks(f) * (rf_block(f) / 64) + 1; }
// The two sweeps of the fused gradient run best on smaller blocks than the 256 threads of the other frequency-lane
// kernels (same-box A/B at config 2: pass A 1.718 -> 1.705 ms at 128, pass B 4.28 -> 4.12 ms at 64): a block is one
// chain's frequencies, and the fewer waves share a block the less a wave waits for its siblings' slots.  The
// partial-sum layout of pass B (one slot per 64-frequency group + the Nyquist slot) does not depend on the block size.
int rf_block_of(const RfFreq& f, int want) { int b = rf_block(f); return b > want ? want : b; }
int rf_chunks_of(const RfFreq& f, int bs) { return (f.n2 - 1 + bs - 1) / bs; }
// pass B's launch covers the frequencies below the band limit (and below the Nyquist bin, which has its own launch and
// its own slot -- only where there is no band limit, it is the first bin to go)
int rf_kmax_b(const RfFreq& f) { return std::min(f.nk, f.n2 - 1); }
int rf_chunks_b(const RfFreq& f) { int bs = rf_block_of(f, 64); return (rf_kmax_b(f) + bs - 1) / bs; }
int rf_nparts_b(const RfFreq& f) { int bs = rf_block_of(f, 64); return rf_chunks_b(f) * (bs / 64) + (f.nk >= f.n2 ? 1 : 0); }

// The band limit of the fused gradient (RfFreq::nk): the adjoint weight of frequency k carries G_k = exp(-(w_k / 2 f0)^2)
// over a denominator >= water * max (RFModule.f90:393,413-419); below eps * water it is lost in the double-precision sum
// over frequencies.  digits = -log10(eps), 0 = no limit.
int band_bins(const RfFreq& f, double digits) {
    const double thr = std::pow(10.0, -digits) * std::min(1.0, f.water);
    const double wcut = 2.0 * f.f0 * std::sqrt(-std::log(thr));           // G(w) >= thr  <=>  w <= wcut
    const double dw = 2.0 * 3.14159265358979323846 / (f.nft * f.dt);
    return (int)std::floor(wcut / dw) + 2;                                // one spare bin: the axis uses a float32 pi
}
// digits: the target (default 13); floor_digits: the least the limit may ever keep (default 8).  A lane is a frequency
// and a wavefront 64 of them, so a limit just above a multiple of 64 leaves a wavefront mostly idle: when a multiple
// of 64 lies between the two limits the band ends there (nt = 512, dt = 0.1, f0 = 1.5: 128 bins = two full wavefronts
// per chain instead of 150 = three; measured gradient difference to the unlimited sum: see DESIGN.md).
void set_band_limit(RfFreq& f, int digits, int floor_digits) {
    f.nk = f.n2; f.nkp = f.n2p;
    if (digits <= 0 || f.method != RFS_RF_FREQ || !(f.water > 0.0) || !(f.f0 > 0.0)) return;
    int nk = band_bins(f, (double)digits);
    if (floor_digits > 0 && floor_digits < digits) {
        const int lo = band_bins(f, (double)floor_digits);
        const int m = (lo + 63) / 64 * 64;
        if (m >= 64 && m <= nk) nk = m;
    }
    if (nk < f.n2) { f.nk = std::max(nk, 1); f.nkp = (f.nk + 15) / 16 * 16; }
}

constexpr int RF_MAX_CHAINS_PER_LAUNCH = 32768;      // the RF sweeps use one grid row per chain (gridDim.y <= 65535)

// pass A (+ scratch) for the nchain chains that start at chain c0 of the batch (RR, Rs are tile-local: offset 0); lc must
// be ready
// peel: pass B will peel the layers off the final row (k_rf_passB<., true>): only that row is kept, no row scratch
// float32 beyond the band: only where there is a band limit (the fused gradient of the frequency-domain method)
bool rf_f32_on(const rfs_ctx* c, const RfFreq& f) { return c->rf_f32 && f.method == RFS_RF_FREQ && f.nk < f.n2 - 64 && f.water > 0.0; }

int launch_passA(rfs_ctx* c, int nchain, int n, const RfFreq& f, bool scratch, size_t c0 = 0, bool peel = false) {
    if (nchain > 2 * RF_MAX_CHAINS_PER_LAUNCH - 4096)
        return fail(c, RFS_ERR_UNSUPPORTED, "more than 61440 chains in one receiver-function launch: split the batch");
    ENSURE(c, c->RR, (size_t)nchain * 4 * f.n2p * sizeof(double));
    // (with peeling the row scratch is still there for the chains that cannot peel -- allocated, untouched by the others)
    if (scratch) ENSURE(c, c->Rs, (size_t)nchain * (n - 1) * 8 * f.nkp * sizeof(double));
    if (scratch && peel) ENSURE(c, c->RT, (size_t)nchain * 8 * f.nkp * sizeof(double));
    double* Rs = scratch ? c->Rs.as<double>() : nullptr;
    double* RT = (scratch && peel) ? c->RT.as<double>() : nullptr;
    int *sl = nullptr, *sc = nullptr;
    if (RT) {
        ENSURE(c, c->slist, (size_t)nchain * sizeof(int));
        if (!c->scount.p) { ENSURE(c, c->scount, 2 * sizeof(int)); HIPCHK(c, hipMemsetAsync(c->scount.p, 0, 2 * sizeof(int), c->stream)); }
        c->speel_eval++;                          // two counters: this evaluation's, and the next one's, which pass A clears
        sl = c->slist.as<int>(); sc = c->scount.as<int>() + (c->speel_eval & 1);
    }
    int* scn = sc ? c->scount.as<int>() + ((c->speel_eval + 1) & 1) : nullptr;
    const RfLayer* lc = c->lc.as<RfLayer>() + c0 * n;
    const int bs = rf_block_of(f, 128);
    dim3 grid(nchain, rf_chunks_of(f, bs) + 1);      // chain = fast index (XCD balance); row 0 = the Nyquist bin (see k_rf_passA)
    RfFreq fa = f;
    int* hi = nullptr;
    if (rf_f32_on(c, f)) { fa.e32max = rf_f32_emax(n); ENSURE(c, c->hi32, (size_t)nchain * sizeof(int)); hi = c->hi32.as<int>(); }
    // "rf_store_hyp": with row peeling pass A leaves the six transcendental numbers of every (layer, band frequency) for pass B
    // (1.5 GB at 8192 chains x 29 layers x 128 frequencies), as long as that fits the scratch budget
    double* Hs = nullptr;
    const size_t hsb = (size_t)nchain * (n - 1) * 6 * f.nkp * sizeof(double);
    if (RT && c->rf_store_hyp && f.nk < f.n2 && hsb <= c->rf_scratch_budget) { ENSURE(c, c->Hs, hsb); Hs = c->Hs.as<double>(); }
    c->Hs_cur = Hs;
    hipLaunchKernelGGL(k_rf_passA, grid, dim3(bs), 0, c->stream, nchain, n, fa, lc, c->RR.as<double>(), Rs, RT, sl, sc, scn, hi, Hs);
    HIPCHK(c, hipGetLastError());
    return RFS_OK;
}

// spectrum -> rf(t): writes dsyn (stride ndata) and optionally misfit/weighted residual
// (mrf: per-chain RF misfits of the whole batch, written at [c0, c0 + nchain); every other buffer is tile-local)
int launch_mid(rfs_ctx* c, int nchain, int n, const RfFreq& f, const double* dobs, int ndata, double* dsyn,
               bool adjoint, size_t c0 = 0, size_t nchain_total = 0) {
    if (adjoint && dobs && c->rf_mid_fused && f.method == RFS_RF_FREQ && f.nft >= 16 && f.nft <= 4096) {
        // the gradient's middle section in one kernel, one chain per block, in LDS (k_rf_mid_fused)
        if (c->twid_nft != f.nft) {
            std::vector<cplx> tw((size_t)f.nft / 2 + 1);
            for (size_t j = 0; j < tw.size(); j++) {
                const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)j / (long double)f.nft;
                tw[j] = C((double)cosl(a), (double)sinl(a));
            }
            HIPCHK(c, hipStreamSynchronize(c->stream));
            ENSURE(c, c->twid, tw.size() * sizeof(cplx));
            HIPCHK(c, hipMemcpy(c->twid.p, tw.data(), tw.size() * sizeof(cplx), hipMemcpyHostToDevice));
            c->twid_nft = f.nft;
            c->mid_tab_key[0] = 0.0;
        }
        const double key[6] = {(double)f.nft, f.dt, f.f0, f.t0, f.sigma, (double)f.nt};
        if (std::memcmp(key, c->mid_tab_key, sizeof(key)) != 0) {
            ENSURE(c, c->gtab, (size_t)f.n2 * sizeof(cplx)); ENSURE(c, c->etab, (size_t)std::max(f.nt, 1) * sizeof(double));
            const int nthr = std::max(f.n2, f.nt);
            hipLaunchKernelGGL(k_rf_mid_tables, dim3((nthr + 255) / 256), dim3(256), 0, c->stream, f, c->gtab.as<cplx>(), c->etab.as<double>());
            HIPCHK(c, hipGetLastError());
            std::memcpy(c->mid_tab_key, key, sizeof(key));
        }
        ENSURE(c, c->wmax2, (size_t)nchain * sizeof(double));
        ENSURE(c, c->mrf, std::max((size_t)nchain, nchain_total) * sizeof(double));
        ENSURE(c, c->W, (size_t)nchain * f.n2 * sizeof(cplx));
        const int* hi = nullptr; unsigned long long* st = nullptr;
        if (rf_f32_on(c, f)) {
            if (!c->stat32.p) { ENSURE(c, c->stat32, 66 * sizeof(unsigned long long)); HIPCHK(c, hipMemsetAsync(c->stat32.p, 0, 66 * sizeof(unsigned long long), c->stream)); }
            hi = c->hi32.as<int>(); st = c->stat32.as<unsigned long long>();
        }
        int logN = 0;
        while ((1 << (logN + 1)) < f.nft) logN++;
        const size_t lds = ((size_t)f.nft / 2 + 1) * sizeof(cplx);
        hipLaunchKernelGGL(k_rf_mid_fused, dim3(nchain), dim3(128), lds, c->stream, n, f, logN, c->lc.as<RfLayer>() + c0 * n,
                           c->RR.as<double>(), c->wmax2.as<double>(), c->twid.as<cplx>(), c->gtab.as<cplx>(), c->etab.as<double>(), dobs, ndata, dsyn,
                           c->mrf.as<double>() + c0, c->W.as<cplx>(), hi, st);
        HIPCHK(c, hipGetLastError());
        return RFS_OK;
    }
    // rocFFT plans are per batch size; varying chain counts (length-sorted trajectories) are rounded up to a
    // multiple of 256 so that a handful of cached plans serve them all (the padding transforms stale data)
    const size_t nb = nchain > 256 ? ((size_t)nchain + 255) / 256 * 256 : (size_t)nchain;
    ENSURE(c, c->wmax2, (size_t)nchain * sizeof(double));
    ENSURE(c, c->spec, nb * f.n2 * sizeof(cplx));
    ENSURE(c, c->tser, nb * f.nft * sizeof(double));
    const int* hi = nullptr; unsigned long long* st = nullptr;
    if (rf_f32_on(c, f)) {
        if (!c->stat32.p) { ENSURE(c, c->stat32, 66 * sizeof(unsigned long long)); HIPCHK(c, hipMemsetAsync(c->stat32.p, 0, 66 * sizeof(unsigned long long), c->stream)); }
        hi = c->hi32.as<int>(); st = c->stat32.as<unsigned long long>();
    }
    hipLaunchKernelGGL(k_rf_mid1, dim3(nchain), dim3(256), 0, c->stream, n, f, c->lc.as<RfLayer>() + c0 * n, c->RR.as<double>(),
                       c->wmax2.as<double>(), c->spec.as<cplx>(), hi, st);
    HIPCHK(c, hipGetLastError());
    TRY(run_fft(c, f.nft, nb, 1, c->spec.p, c->tser.p));
    double* wres = nullptr; double* mrf = nullptr;
    if (adjoint) {
        ENSURE(c, c->wres, nb * f.nft * sizeof(double));
        ENSURE(c, c->mrf, std::max((size_t)nchain, nchain_total) * sizeof(double));
        ENSURE(c, c->W, nb * f.n2 * sizeof(cplx));
        wres = c->wres.as<double>(); mrf = c->mrf.as<double>() + c0;
    }
    hipLaunchKernelGGL(k_rf_mid2, dim3(nchain), dim3(256), 0, c->stream, f, c->tser.as<double>(), dobs, ndata,
                       dsyn, mrf, wres);
    HIPCHK(c, hipGetLastError());
    if (adjoint) TRY(run_fft(c, f.nft, nb, 0, c->wres.p, c->W.p));
    return RFS_OK;
}

int launch_passB(rfs_ctx* c, int nchain, int n, const RfFreq& f, size_t c0 = 0, bool peel = false) {
    int npart = rf_nparts_b(f);
    ENSURE(c, c->PG, (size_t)nchain * npart * 4 * n * sizeof(double));
    const RfLayer* lc = c->lc.as<RfLayer>() + c0 * n;
    const int bs = rf_block_of(f, 64);
    dim3 grid(rf_chunks_b(f), nchain);
    const double* rows = c->Rs.as<double>();
    const double* rt = peel ? c->RT.as<double>() : (const double*)nullptr;
    if (peel && !c->rstat.p) { ENSURE(c, c->rstat, sizeof(unsigned)); HIPCHK(c, hipMemsetAsync(c->rstat.p, 0, sizeof(unsigned), c->stream)); }
    unsigned* pr = c->rf_peel_check ? c->rstat.as<unsigned>() : nullptr;
    // with peeling: one launch per kind of chain.  The stored-row one strides over the list pass A made of its chains --
    // normally empty, so its grid follows the length seen in an earlier evaluation (never waited for) and a handful of blocks
    // find nothing to do; whatever the list holds is processed, only slower while the estimate lags
    const int* nol = nullptr; int* noe = nullptr;
    if (peel) {
        if (!c->h_scount) {
            HIPCHK(c, hipHostMalloc((void**)&c->h_scount, sizeof(int), hipHostMallocMapped));
            *c->h_scount = -1;
            HIPCHK(c, hipHostGetDevicePointer((void**)&c->d_hscount, c->h_scount, 0));
        }
        if (*c->h_scount >= 0) c->stored_est = *c->h_scount;
        if (c->Hs_cur)
            hipLaunchKernelGGL((k_rf_passB<false, true, true>), grid, dim3(bs), 0, c->stream, nchain, n, f, lc, c->RR.as<double>(), rows, rt,
                               c->W.as<cplx>(), c->wmax2.as<double>(), npart, c->PG.as<double>(), pr, nol, nol, noe, (const double*)c->Hs_cur);
        else
            hipLaunchKernelGGL((k_rf_passB<false, true>), grid, dim3(bs), 0, c->stream, nchain, n, f, lc, c->RR.as<double>(), rows, rt,
                               c->W.as<cplx>(), c->wmax2.as<double>(), npart, c->PG.as<double>(), pr, nol, nol, noe, (const double*)nullptr);
        const int gy = c->stored_est < 0 ? nchain : std::max(8, std::min(nchain, c->stored_est + c->stored_est / 8));
        hipLaunchKernelGGL((k_rf_passB<false, false>), dim3(rf_chunks_b(f), gy), dim3(bs), 0, c->stream, nchain, n, f, lc,
                           c->RR.as<double>(), rows, rt, c->W.as<cplx>(), c->wmax2.as<double>(), npart, c->PG.as<double>(), pr,
                           c->slist.as<int>(), c->scount.as<int>() + (c->speel_eval & 1), c->d_hscount);
    } else {
        hipLaunchKernelGGL((k_rf_passB<false, false>), grid, dim3(bs), 0, c->stream, nchain, n, f, lc, c->RR.as<double>(), rows, rt,
                           c->W.as<cplx>(), c->wmax2.as<double>(), npart, c->PG.as<double>(), pr, nol, nol, noe);
    }
    if (f.nk >= f.n2) {     // the Nyquist bin, lane = chain (no band limit: it is the first bin to go)
        if (peel) hipLaunchKernelGGL((k_rf_passB<true, true>), dim3((nchain + 63) / 64), dim3(64), 0, c->stream, nchain, n, f, lc,
                                     c->RR.as<double>(), rows, rt, c->W.as<cplx>(), c->wmax2.as<double>(), npart, c->PG.as<double>(), pr, nol, nol, noe);
        hipLaunchKernelGGL((k_rf_passB<true, false>), dim3((nchain + 63) / 64), dim3(64), 0, c->stream, nchain, n, f, lc,
                           c->RR.as<double>(), rows, rt, c->W.as<cplx>(), c->wmax2.as<double>(), npart, c->PG.as<double>(), pr, nol, nol, noe);
    }
    HIPCHK(c, hipGetLastError());
    return RFS_OK;
}

// ------------------------------------------------------------------ time-domain RF (rf_time_kernels.hpp)
int rft_pulse(rfs_ctx* c, const RfFreq& f) {     // gauss + shift pulse of deconit's tail, cached per (nft, dt, f0, t0)
    double key[4] = {(double)f.nft, f.dt, f.f0, f.t0};
    if (c->pulse_ts.p && std::memcmp(key, c->pulse_key, sizeof(key)) == 0) return RFS_OK;
    ENSURE(c, c->pulse_spec, (size_t)f.n2 * sizeof(cplx));
    ENSURE(c, c->pulse_ts, (size_t)f.nft * sizeof(double));
    hipLaunchKernelGGL(k_rft_pulse_spec, dim3((f.n2 + 255) / 256), dim3(256), 0, c->stream, f, c->pulse_spec.as<cplx>());
    HIPCHK(c, hipGetLastError());
    TRY(run_fft(c, f.nft, 1, 1, c->pulse_spec.p, c->pulse_ts.p));
    std::memcpy(c->pulse_key, key, sizeof(key));
    return RFS_OK;
}

template <int WPB>
int rft_launch_deconv(rfs_ctx* c, int ntrace, int tpc, const RfFreq& f, const double* cuw0, size_t cuw_stride,
                      const double* aw, size_t aw_stride, const double* S0, int nS0, size_t s0c, size_t s0p,
                      const double* Cres, double* Pout, double* gout) {
    if (f.nft > 4096) {                  // one block per trace, lags in place (k_rft_deconv_big)
        const int bt = (int)std::min<size_t>(1024, (size_t)f.nft / 2);
        if (Pout) hipLaunchKernelGGL(k_rft_deconv_big<true>, dim3(ntrace), dim3(bt), 0, c->stream, ntrace, tpc, f, const_cast<double*>(cuw0),
                                     cuw_stride, aw, aw_stride, S0, nS0, s0c, s0p, Cres, Pout, gout);
        else hipLaunchKernelGGL(k_rft_deconv_big<false>, dim3(ntrace), dim3(bt), 0, c->stream, ntrace, tpc, f, const_cast<double*>(cuw0),
                                cuw_stride, aw, aw_stride, S0, nS0, s0c, s0p, Cres, Pout, gout);
        HIPCHK(c, hipGetLastError());
        return RFS_OK;
    }
    int npl = f.nft / 128; if (npl < 1) npl = 1;
    dim3 grid((ntrace + WPB - 1) / WPB), block(64 * WPB);
    size_t lds = ((size_t)2 * f.nft + (Cres ? f.nft / 2 : 0)) * sizeof(double);
#define RFS_DECONV2(NPL, FULL, WANTP)                                                                            \
    hipLaunchKernelGGL((k_rft_deconv<NPL, WPB, FULL, WANTP>), grid, block, lds, c->stream, ntrace, tpc, f, cuw0, \
                       cuw_stride, aw, aw_stride, S0, nS0, s0c, s0p, Cres, Pout, gout, (int*)nullptr)
#define RFS_DECONV(NPL, FULL) do { if (Pout) RFS_DECONV2(NPL, FULL, true); else RFS_DECONV2(NPL, FULL, false); } while (0)
    if (lds > 64 * 1024) {
        HIPCHK(c, hipFuncSetAttribute((const void*)k_rft_deconv<32, WPB, true, true>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(c, hipFuncSetAttribute((const void*)k_rft_deconv<32, WPB, true, false>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    switch (npl) {
        case 1: if (f.nft >= 128) RFS_DECONV(1, true); else RFS_DECONV(1, false); break;
        case 2: RFS_DECONV(2, true); break;
        case 4: RFS_DECONV(4, true); break;
        case 8: RFS_DECONV(8, true); break;
        case 16: RFS_DECONV(16, true); break;
        default: RFS_DECONV(32, true); break;
    }
#undef RFS_DECONV2
#undef RFS_DECONV
    HIPCHK(c, hipGetLastError());
    return RFS_OK;
}

// after pass A: chain-level spectra + their inverse transforms, forward-trace deconvolution, rf(t) -> out
int rft_forward(rfs_ctx* c, int nchain, const RfFreq& f, double* out, size_t ostride) {
    const size_t nft = f.nft, half = nft / 2;
    const size_t nb = nchain > 256 ? ((size_t)nchain + 255) / 256 * 256 : (size_t)nchain;    // see launch_mid
    ENSURE(c, c->spec3, nb * 3 * f.n2 * sizeof(cplx));
    ENSURE(c, c->ts3, nb * 3 * nft * sizeof(double));
    ENSURE(c, c->S0f, (size_t)nchain * sizeof(double));
    ENSURE(c, c->Pbuf, (size_t)nchain * half * sizeof(double));
    TRY(rft_pulse(c, f));
    hipLaunchKernelGGL(k_rft_chain_spectra, dim3(nchain), dim3(256), 0, c->stream, f, c->RR.as<double>(),
                       c->spec3.as<cplx>(), c->S0f.as<double>());
    HIPCHK(c, hipGetLastError());
    TRY(run_fft(c, f.nft, nb * 3, 1, c->spec3.p, c->ts3.p));
    TRY(rft_launch_deconv<1>(c, nchain, 1, f, c->ts3.as<double>() + nft, 3 * nft, c->ts3.as<double>(), 3 * nft,
                             c->S0f.as<double>(), 1, 1, 0, nullptr, c->Pbuf.as<double>(), nullptr));
    size_t lds = (nft + half + half / 2 + 1) * sizeof(double);
    if (nft > 4096) hipLaunchKernelGGL(k_rft_synth_big, dim3(nchain), dim3(256), 0, c->stream, nchain, f, c->Pbuf.as<double>(),
                                       c->pulse_ts.as<double>(), out, ostride);
    else
    hipLaunchKernelGGL(k_rft_synth, dim3(nchain), dim3(256), lds, c->stream, nchain, f, c->Pbuf.as<double>(),
                       c->pulse_ts.as<double>(), out, ostride);
    HIPCHK(c, hipGetLastError());
    return RFS_OK;
}

// the 4n partial traces of every chain (after rft_forward): either their spike trains -> kl[chain][4][n][nt]
// (B1 kernel_all) or, with Cres given, directly sum_t k(t) r(t) -> PG[chain][4][n] (B2 gradient).  Chains are
// processed in chunks so that spectra + time series stay below ~6 GB.
int rft_partials(rfs_ctx* c, int nchain, int n, const RfFreq& f, const double* Cres, double* PG, double* kl) {
    const size_t nft = f.nft, half = nft / 2, ntr = (size_t)4 * n;
    const int npart = rf_nparts(f);
    size_t per_chain = ntr * (f.n2 * sizeof(cplx) + nft * sizeof(double));
    int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)nchain, (size_t)6e9 / per_chain));
    ENSURE(c, c->specp, (size_t)chunk * ntr * f.n2 * sizeof(cplx));
    ENSURE(c, c->tserp, (size_t)chunk * ntr * nft * sizeof(double));
    ENSURE(c, c->S0p, (size_t)chunk * npart * ntr * sizeof(double));
    if (kl) ENSURE(c, c->Pbuf, std::max((size_t)nchain * half, (size_t)chunk * ntr * half) * sizeof(double));
    for (int c0 = 0; c0 < nchain; c0 += chunk) {
        int nc = std::min(chunk, nchain - c0);
        dim3 grid(rf_chunks(f), nc);
        hipLaunchKernelGGL(k_rft_partial_spectra<false>, grid, dim3(rf_block(f)), 0, c->stream, nc, n, f,
                           c->lc.as<RfLayer>() + (size_t)c0 * n, c->RR.as<double>() + (size_t)c0 * 4 * f.n2p,
                           c->Rs.as<double>() + (size_t)c0 * (n - 1) * 8 * f.nkp, npart, c->specp.as<cplx>(),
                           c->S0p.as<double>());
        hipLaunchKernelGGL(k_rft_partial_spectra<true>, dim3((nc + 63) / 64), dim3(64), 0, c->stream, nc, n, f,
                           c->lc.as<RfLayer>() + (size_t)c0 * n, c->RR.as<double>() + (size_t)c0 * 4 * f.n2p,
                           c->Rs.as<double>() + (size_t)c0 * (n - 1) * 8 * f.nkp, npart, c->specp.as<cplx>(),
                           c->S0p.as<double>());
        HIPCHK(c, hipGetLastError());
        TRY(run_fft(c, f.nft, (size_t)nc * ntr, 1, c->specp.p, c->tserp.p));
        TRY(rft_launch_deconv<4>(c, (int)(nc * ntr), (int)ntr, f, c->tserp.as<double>(), nft,
                                 c->ts3.as<double>() + (size_t)c0 * 3 * nft + 2 * nft, 3 * nft, c->S0p.as<double>(),
                                 npart, (size_t)npart * ntr, ntr, Cres ? Cres + (size_t)c0 * half : nullptr,
                                 kl ? c->Pbuf.as<double>() : nullptr, PG ? PG + (size_t)c0 * ntr : nullptr));
        if (kl) {
            size_t lds = (nft + half + half / 2 + 1) * sizeof(double);
            if (nft > 4096) hipLaunchKernelGGL(k_rft_synth_big, dim3((unsigned)(nc * ntr)), dim3(256), 0, c->stream, (int)(nc * ntr), f,
                               c->Pbuf.as<double>(), c->pulse_ts.as<double>(), kl + (size_t)c0 * ntr * f.nt, (size_t)f.nt);
            else
            hipLaunchKernelGGL(k_rft_synth, dim3((unsigned)(nc * ntr)), dim3(256), lds, c->stream, (int)(nc * ntr), f,
                               c->Pbuf.as<double>(), c->pulse_ts.as<double>(), kl + (size_t)c0 * ntr * f.nt, (size_t)f.nt);
            HIPCHK(c, hipGetLastError());
        }
    }
    return RFS_OK;
}

// Sequences and data rows of one evaluation.  Rayleigh sequences come first (seq 0..), then Love; items
// (periods of all sequences) are numbered in the same order, so both families share croot / krn / ugr.
// group_passes: group blocks get the three root searches T, 1.05 T, 0.95 T of surfdisp.cpp:235-241;
// otherwise (forward) only T.  love_group_vp: Lg searches with vp = 1.732 vs (_LoveGroup, surfdisp.cpp:132).
struct SwdPlan {
    SwdSeqs QR{}, QL{};
    SwdRows R{};
    int nseq = 0, nitems = 0;
    bool any_love() const { return QL.nseq > 0; }
};

SwdPlan make_plan(const int nt[4], const double* const t[4], bool group_passes, int sphere, int fwd,
                  bool love_group_vp, const double* sphR, const double* sphL, bool alias_rg = false, bool alias_lg = false) {
    SwdPlan P;
    int off = 0;
    int boff[4] = {0, 0, 0, 0}, boff1[4] = {0, 0, 0, 0}, boff2[4] = {0, 0, 0, 0};
    for (int type = 0; type < 4; type++) {
        if (nt[type] <= 0) continue;
        SwdSeqs& Q = (type < 2) ? P.QR : P.QL;
        int alt = (type == 3 && love_group_vp) ? 1 : 0;
        // alias_rg: the Rg block has the Rc block's periods.  Its pass at T (sregnpu's central pass, surfdisp.cpp:235-241)
        // is then the very search and the very eigenfunction pass of the Rc block: it reads those items instead of
        // repeating them (param.yaml's own set-up, tRc = tRg, saves a quarter of the Rayleigh work that way)
        // (the same for Lg / Lc, except in libsurf.forward's form, where _LoveGroup searches with its own P velocity)
        if (type == 1 && alias_rg && nt[0] == nt[1] && nt[0] > 0) {
            boff[type] = boff[0];
        } else if (type == 3 && alias_lg && !love_group_vp && nt[2] == nt[3] && nt[2] > 0) {
            boff[type] = boff[2];
        } else {
            boff[type] = off;
            Q.s[Q.nseq++] = SwdSeq{t[type], nt[type], 1.0, off, alt}; off += nt[type];
            Q.nper_total += nt[type];
        }
        if ((type & 1) && group_passes) {
            boff1[type] = off;
            Q.s[Q.nseq++] = SwdSeq{t[type], nt[type], 1.0 + 0.05, off, alt}; off += nt[type];
            boff2[type] = off;
            Q.s[Q.nseq++] = SwdSeq{t[type], nt[type], 1.0 - 0.05, off, alt}; off += nt[type];
            Q.nper_total += 2 * nt[type];
        }
    }
    P.nseq = P.QR.nseq + P.QL.nseq; P.nitems = off;
    for (int type = 0; type < 4; type++) {
        if (nt[type] <= 0) continue;
        SwdBlk B{type, nt[type], boff[type], boff1[type], boff2[type], t[type]};
        P.R.b[P.R.nblk++] = B;
        P.R.nswd += nt[type];
    }
    P.R.sphere = sphere; P.R.fwd = fwd; P.R.sphR = sphR; P.R.sphL = sphL; P.R.nitems = P.nitems;
    return P;
}

// Launch shape of the cooperative root search (k_swd_roots_coop<NCH>): 512-thread blocks of 64 (sequence, chain) items,
// one block per CU (107 KB of LDS)
// Up to this many (sequence, chain) items the device is mostly idle and the root search is a pure latency problem: the
// lanes-per-item kernel with segmented recurrence and scan look-ahead (k_swd_roots_split) beats the cooperative blocks
// (measured at 30 layers: 1 item 8.0 -> 3.0 ms, 1024 items 6.8 -> 4.3 ms, 3072 items 7.0 -> 6.2 ms, level at ~3500 items).
constexpr int SWD_LAT_MAX_ITEMS = 3072;
struct CoopPlan { bool ok = false; int nch = 0, blocks = 0, per_cu = 1; size_t lds = 0; };

CoopPlan coop_plan(const rfs_ctx* c, const SwdSeqs& Q, int nchain, int n) {
    CoopPlan P;
    const int nitem = Q.nseq * nchain;
    int npmax = 0;
    for (int q = 0; q < Q.nseq; q++) npmax = Q.s[q].nper > npmax ? Q.s[q].nper : npmax;
    if (c->swd_lanes != 0 || nitem <= SWD_LAT_MAX_ITEMS || n < 3 || n - 2 > 16 * COOP_NP || Q.nseq * npmax > 4096) return P;
    P.nch = (n - 1 - COOP_CL + COOP_NP - 1) / COOP_NP;
    P.lds = (size_t)(4 * 64 + 8 + 2 * COOP_NP * SWD_NENT * 64 + 24 * 64 + 2 * Q.nseq * npmax) * sizeof(double);
    P.blocks = (nitem + 63) / 64;
    P.ok = true;
    return P;
}

// per-family search models (earth flattening, Love); mdl must be ready on stream s
int launch_family_prep(rfs_ctx* c, hipStream_t s, int nchain, int n, const SwdPlan& P, int sphere) {
    const bool wantR = P.QR.nseq > 0, wantL = P.QL.nseq > 0;
    if (!(sphere && wantR) && !wantL) return RFS_OK;
    size_t nn = (size_t)n * nchain;
    if (sphere && wantR) { ENSURE(c, c->mdlSR, 4 * nn * sizeof(float)); ENSURE(c, c->sphR, 7 * nn * sizeof(double)); }
    if (wantL) { ENSURE(c, c->mdlL, 5 * nn * sizeof(float)); ENSURE(c, c->mdlcL, 6 * nn * sizeof(double)); if (sphere) ENSURE(c, c->sphL, 7 * nn * sizeof(double)); }
    hipLaunchKernelGGL(k_prep_swd_family, dim3((nchain + 63) / 64), dim3(64), 0, s, nchain, n, c->mdl.as<float>(), sphere,
                       (int)wantR, (int)wantL, c->mdlSR.as<float>(), c->mdlc.as<double>(), c->sphR.as<double>(),
                       c->mdlL.as<float>(), c->sphL.as<double>(), c->mdlcL.as<double>());
    HIPCHK(c, hipGetLastError());
    return RFS_OK;
}

// The lanes-per-item root search (k_swd_roots_split) for one wave family.  coop_ok: the cooperative blocks take the big
// Rayleigh batches, so G is only needed up to SWD_LAT_MAX_ITEMS there.
template <class F>
int launch_roots_split(rfs_ctx* c, hipStream_t s, int nchain, int n, const SwdSeqs& Q, const float* mdl, const double* mdlc,
                       int* sflag, int G, const int* list = nullptr, const int* count = nullptr, int est_chains = 0) {
    // list: the search of the chains a warm start handed back -- their number is only known on the device, the launch
    // shape follows the host's estimate and the blocks stride over whatever the list holds
    const int nitem = list ? Q.nseq * std::max(1, std::min(est_chains, nchain)) : Q.nseq * nchain;
    int spec_auto = 1;
    if (G <= 0 && nitem <= SWD_LAT_MAX_ITEMS) {
        // latency mode: one item per wavefront while the device has room for it (no two items' state machines diverging
        // inside a wavefront), then two, then four; 4 / 2 wavefronts per block look ahead in the scan
        if (nitem <= 256) { G = 64; spec_auto = 4; }
        else if (nitem <= 1152) { G = 32; spec_auto = 4; }
        else { G = 16; spec_auto = 2; }
    } else if (G <= 0) {    // about one wave per SIMD (1024 of them): G = 65536 / items, within [4, 32]
        G = 4;
        while (G < 32 && (size_t)nitem * G * 2 <= 65536 && 2 * G <= n - 1) G *= 2;
    }
    size_t lds = (size_t)(n - 1) * F::NENT * (64 / G) * sizeof(double);
    while (G > 1 && G < 64 && (lds > 60 * 1024 || (n - 1 + G - 1) / G > 8)) {
        G *= 2; lds = (size_t)(n - 1) * F::NENT * (64 / G) * sizeof(double);
    }
    if (G == 1) return 1;                    // caller falls back to the lane-per-item kernel
    int NG = 64 / G, lpl = (n - 1 + G - 1) / G;
    if (lpl > 8 || lds > 60 * 1024) return 1;   // more than 8 layers per lane even with 64 lanes (> 513 layers): same fallback
    dim3 grid(list ? std::min((Q.nseq * nchain + NG - 1) / NG, std::max(64, 2 * (nitem + NG - 1) / NG)) : (nitem + NG - 1) / NG);
    // few items (the device is mostly idle): cut the vector recurrence into segments that run side by side on the
    // group's lanes -- 4 segments need 1 + 3 NV lanes, 2 need 1 + NV
    int nseg = c->swd_segments;
    if (nseg < 0) nseg = (nitem <= SWD_LAT_MAX_ITEMS) ? (G >= 16 && n - 1 >= 8 ? 4 : (G >= 8 && n - 1 >= 4 ? 2 : 1)) : 1;
    if ((nseg == 4 && G < 16) || (nseg == 2 && G < 8) || (nseg != 2 && nseg != 4)) nseg = 1;
    // ... and let 4 wavefronts per block look ahead in the scan
    int spec = c->swd_speculate;
    if (spec < 0) spec = spec_auto;
    if (spec != 2 && spec != 4) spec = 1;
    auto lds_of = [&](int sp) {
        return (size_t)sp * (lds + (size_t)(2 + F::NV * (nseg - 1)) * F::NV * NG * sizeof(double)) +
               (sp > 1 ? (size_t)2 * sp * NG * sizeof(double) : 0);
    };
    while (spec > 1 && lds_of(spec) > 60 * 1024) spec /= 2;
    size_t lds_s = lds_of(spec);
#define RFS_LAUNCH_SPLIT3(LPL, NSEG, SPEC)                                                                          \
    hipLaunchKernelGGL((k_swd_roots_split<F, LPL, NSEG, SPEC>), grid, dim3(64 * SPEC), lds_s, s, nchain, n, G, Q,   \
                       mdl, mdlc, c->croot.as<double>(), sflag, list, count)
#define RFS_LAUNCH_SPLIT2(LPL, NSEG)                                                                                \
    do { if (spec == 4) RFS_LAUNCH_SPLIT3(LPL, NSEG, 4); else if (spec == 2) RFS_LAUNCH_SPLIT3(LPL, NSEG, 2);       \
         else RFS_LAUNCH_SPLIT3(LPL, NSEG, 1); } while (0)
#define RFS_LAUNCH_SPLIT(LPL)                                                                                       \
    do { if (nseg == 4) RFS_LAUNCH_SPLIT2(LPL, 4); else if (nseg == 2) RFS_LAUNCH_SPLIT2(LPL, 2);                   \
         else RFS_LAUNCH_SPLIT2(LPL, 1); } while (0)
    if (lpl <= 1) RFS_LAUNCH_SPLIT(1);
    else if (lpl <= 2) RFS_LAUNCH_SPLIT(2);
    else if (lpl <= 4) RFS_LAUNCH_SPLIT(4);
    else RFS_LAUNCH_SPLIT(8);
#undef RFS_LAUNCH_SPLIT
#undef RFS_LAUNCH_SPLIT2
#undef RFS_LAUNCH_SPLIT3
    return 0;
}

// Cooperative producer / consumer blocks for the Love family (big batches; the Rayleigh launch below has its own plan,
// which the step's CU partition is built around).  Returns 1 when the shape does not fit (caller falls back).
int launch_love_coop(rfs_ctx* c, hipStream_t s, int nchain, int n, const SwdSeqs& Q, const float* mdl, const double* mdlc,
                     int* sflag, const int* list = nullptr, const int* count = nullptr, int est_chains = 0) {
    const int nitem = list ? Q.nseq * std::max(1, std::min(est_chains, nchain)) : Q.nseq * nchain;
    int npmax = 0;
    for (int q = 0; q < Q.nseq; q++) npmax = Q.s[q].nper > npmax ? Q.s[q].nper : npmax;
    if (c->swd_lanes != 0 || nitem <= SWD_LAT_MAX_ITEMS || n < 3 || n - 2 > 16 * COOP_NP || Q.nseq * npmax > 4096) return 1;
    const int nch = (n - 1 - COOP_CL + COOP_NP - 1) / COOP_NP;
    const size_t lds = (size_t)(4 * 64 + 8 + 2 * COOP_NP * SwdLoveFamily::NENT * 64 + 24 * 64 + 2 * Q.nseq * npmax) * sizeof(double);
    if (lds > 60 * 1024) return 1;
    dim3 grid((Q.nseq * nchain + 63) / 64);      // (with a list: sized for every chain, blocks beyond the list leave at once)
    if (!list) grid = dim3((nitem + 63) / 64);
#define RFS_LAUNCH_LCOOP(NCH)                                                                                  \
    hipLaunchKernelGGL((k_swd_roots_coop<SwdLoveFamily, NCH>), grid, dim3(512), lds, s, nchain, n, Q, mdl, mdlc, \
                       c->croot.as<double>(), sflag, list, count)
    if (nch <= 5) RFS_LAUNCH_LCOOP(5);
    else if (nch <= 8) RFS_LAUNCH_LCOOP(8);
    else RFS_LAUNCH_LCOOP(16);
#undef RFS_LAUNCH_LCOOP
    return 0;
}

// root search (+ eigenfunction kernels) on stream `s`; mdl must be ready
// eigen_mode 0: every item; 3 / 4: the Rayleigh / the Love items only; 1: EARLY launch of the Rayleigh items [0, early_items) beside a running search;
// 2: MOP-UP of what the early launch left (k_swd_eigen)
int launch_swd(rfs_ctx* c, hipStream_t s, int nchain, int n, const SwdPlan& P, bool kernels, bool roots = true,
               int eigen_mode = 0, int early_items = 0, bool warm = false) {
    const SwdSeqs& Q = P.QR;
    const int sphere = P.R.sphere;
    ENSURE(c, c->croot, (size_t)P.nitems * nchain * sizeof(double));
    ENSURE(c, c->sflag, (size_t)8 * nchain * sizeof(int));
    const float* mdlR = sphere ? c->mdlSR.as<float>() : c->mdl.as<float>();
    int* sflagL = c->sflag.as<int>() + (size_t)P.QR.nseq * nchain;
    hipStream_t warm_side = nullptr;         // side stream carrying the full search of the chains a warm start handed back
    bool bg_record = false;                  // this step's searches of handed-back chains stay in the background
    bool walk_join = false;                  // the grid walk ran on its own stream: this stream joins it behind the eigenfunction pass
    if (!(warm && roots)) {
        if (roots) c->need_prev_par = -1;     // (the next warm-started evaluation has no warm-started one before it)
        // whatever comes now rewrites the root buffer: every background search still under way has to be through
        for (int i = 0; i < RFS_BG_SLOTS; i++)
            if (c->bg_busy[i]) { HIPCHK(c, hipStreamWaitEvent(s, c->ev_bg[i], 0)); }
    }
    if (warm && roots) {
        // Inside a trajectory: every (period, chain) item refines the previous evaluation's root on its own (k_swd_warm);
        // the chains that cannot be continued go through the reference-semantics search right behind, on a list.
        KTimer* tw = new KTimer(c, RFS_K_SWD_ROOTS, s);      // (closed in front of the reference-root stage, which is a group of its own)
        struct TwGuard { KTimer*& t; ~TwGuard() { delete t; t = nullptr; } } tw_guard{tw};
        // (two sets of flags / lists, alternating: a background search of the step before may still be reading the other one)
        int* wn = c->wneed.as<int>() + (size_t)c->wpar * (3 * (size_t)nchain + 4);
        const size_t lo = (size_t)c->wpar * nchain;
        SwdWarm W{c->dxT.as<double>(), c->wvalid.as<int>(), c->exact_final ? c->wforce.as<int>() : (const int*)nullptr,
                  wn, wn + nchain, c->wlist.as<int>() + lo, c->wstats.as<unsigned long long>(),
                  c->wsgn.as<unsigned char>(), wn + nchain + 1, wn + 2 * nchain + 1,
                  c->wilist.as<int>(), wn + 2 * nchain + 2, c->wlist2.as<int>() + lo, wn + 2 * nchain + 3,
                  c->wslope.as<double>(), c->wbetmx.as<float>(), c->warm_exact ? c->cwarm.as<double>() : (double*)nullptr,
                  wn + 3 * nchain + 3, c->wlist3.as<int>() + lo,
                  (c->flow_cur && c->fpend.p && c->fpend.cap >= (size_t)nchain * sizeof(int)) ? c->fpend.as<int>() : (const int*)nullptr,
                  c->wsg1.as<unsigned char>(),
                  (c->flow_cur && c->flow_skip_idle) ? c->f_rem : (const int*)nullptr, c->f_fresh, c->f_ok,
                  c->warm_exact ? c->cwarm.as<double>() : c->croot.as<double>(), c->warm_widen ? 1 : 0,
                  c->warm_feedback ? c->wferr.as<double>() : (double*)nullptr, c->walk_window, 0, (const int*)nullptr, (int*)nullptr};
        (void)0;
// Rounds (WarmSpill, rfsurf_kernels.hpp): budgets b1, b2, b3 and a last round without one; the unfinished searches of a round
        // are packed into a list for the next.  The lists' lengths are only known on the device: the later rounds' grids are sized for
        // what the bench's chains need several times over, and their blocks stride.
        // Small batches whose hand-backs are searched in the foreground (round 6, "swd_cold_scan"): ONE list for the three kinds of
        // hand-back; the chains the warm search itself declines go through the search without a prediction (k_swd_cold_scan /
        // k_swd_cold_pick) and on to the branch test with everybody else; what is on the list behind the reference-root stage is
        // searched sequentially on this very stream, in front of the eigenfunction pass of all chains -- 12 launches on the step's
        // chain instead of 19 (three searches and three eigenfunction passes over lists that are empty most of the time).
        const bool would_async = c->flow_cur && c->flow_async && kernels && c->stream_l && s != c->stream_l;
        const int np_cold = std::max(Q.nper_total, P.QL.nper_total);
        int npseq_cold = 0;
        for (int q = 0; q < Q.nseq; q++) npseq_cold = std::max(npseq_cold, Q.s[q].nper);
        for (int q = 0; q < P.QL.nseq; q++) npseq_cold = std::max(npseq_cold, P.QL.s[q].nper);
        const bool sb = !would_async && c->cold_scan != 0 && n >= 3 && c->swd_mode_cur == 0 && !c->swd_water_cur &&
                        nchain <= (c->cold_scan > 0 ? COLD_MAX_CHAINS : COLD_AUTO_CHAINS) && npseq_cold <= 192;     // (k_swd_cold_pick holds a sequence's roots in LDS: 268 B a period)
        const int wb1 = c->warm_budgets % 100, wb2 = sb ? 0 : (c->warm_budgets / 100) % 100, wb3 = (c->warm_budgets / 10000) % 100;
        const bool rounds = wb1 > 0;
        if (sb) { W.list2 = W.list3 = W.list; W.count2 = W.count3 = W.count; }
        const bool cold_first = sb && nchain <= c->cold_first;
        W.decline_all = cold_first ? 1 : 0;
        // (the hand-back flags of the evaluation before this one, if that was a warm-started evaluation of the same chains)
        if (sb && !cold_first && c->cold_again) {
            const size_t before = c->wcold.cap;
            ENSURE(c, c->wcold, (size_t)nchain * sizeof(int));
            if (c->wcold.cap != before) HIPCHK(c, hipMemsetAsync(c->wcold.p, 0, c->wcold.cap, s));
            W.cold_again = c->wcold.as<int>();
            if (c->need_prev_par >= 0 && c->need_prev_nchain == nchain)
                W.need_prev = c->wneed.as<int>() + (size_t)c->need_prev_par * (3 * (size_t)nchain + 4);
        }
        c->need_prev_par = c->wpar; c->need_prev_nchain = nchain;
        const size_t items_max = (size_t)std::max(Q.nper_total, P.QL.nper_total) * nchain;
        // (measured need: a few per cent of the items; a list that overflows is not an error -- the searches it cannot take finish
        // in place, test_warm_search_in_rounds_... runs that path -- so a quarter of the items is plenty: 152 B a slot, two lists)
        const int spcap = rounds ? (int)std::min<size_t>(std::max<size_t>(4096, items_max / 4), (size_t)1 << 30) : 1;
        const size_t spbytes = (size_t)spcap * (WARM_SPILL_ND + 2) * 8;
        ENSURE(c, c->wspA, spbytes); ENSURE(c, c->wspB, spbytes); ENSURE(c, c->wspc, 8 * sizeof(int));
        auto spill = [&](Buf& bf, int ci) {
            double* d = bf.as<double>();
            return WarmSpill{d, (unsigned long long*)(d + (size_t)WARM_SPILL_ND * spcap), (unsigned long long*)(d + (size_t)(WARM_SPILL_ND + 1) * spcap),
                             c->wspc.as<int>() + ci, spcap};
        };
        const int NOLIM = 0x7fffffff;
#define RFS_LAUNCH_WARM1(FAM, SPHB, FIRSTB, GRID, QQ, MDLC, SPHP, IN, OUT, BUD, RND)                                   \
        hipLaunchKernelGGL((k_swd_warm<FAM, SPHB, FIRSTB>), GRID, dim3(64), 0, s, nchain, n, QQ, MDLC, SPHP,             \
                           c->krn.as<double>(), c->ugr.as<double>(), (size_t)P.nitems * nchain, c->croot.as<double>(), W, \
                           IN, OUT, BUD, RND)
        // the last round with 16 lanes per search (k_swd_warm_coop): its grid strides as well -- 4 searches per wavefront
        const bool coop_last = rounds && c->warm_coop && n >= 3 && n - 1 <= 64;
#define RFS_LAUNCH_WARMC(FAM, QQ, MDLC, IN)                                                                            \
        do {                                                                                                          \
            const size_t ldsb = (size_t)4 * ((size_t)(n - 1) * FAM::NENT + FAM::NV) * sizeof(double);                 \
            const unsigned gc = (unsigned)std::min<size_t>(((size_t)(QQ).nper_total * nchain + 3) / 4, (size_t)8192); \
            hipLaunchKernelGGL((k_swd_warm_coop<FAM>), dim3(std::max(64u, gc)), dim3(64), ldsb, s, nchain, n, QQ, MDLC, \
                               c->croot.as<double>(), W, IN);                                                         \
        } while (0)
#define RFS_LAUNCH_WARM(FAM, QQ, MDLC, SPHP)                                                                          \
        do {                                                                                                          \
            const size_t nit = (size_t)(QQ).nper_total * nchain;                                                      \
            dim3 grid((unsigned)((nit + 63) / 64));                                                                   \
            const double* sp_ = sphere ? (SPHP) : (const double*)nullptr;                                             \
            WarmSpill none{nullptr, nullptr, nullptr, c->wspc.as<int>() + 3, 0};                                      \
            if (!rounds || cold_first) {                                                                              \
                if (sphere) RFS_LAUNCH_WARM1(FAM, true, true, grid, QQ, MDLC, sp_, none, none, NOLIM, -1);            \
                else RFS_LAUNCH_WARM1(FAM, false, true, grid, QQ, MDLC, sp_, none, none, NOLIM, -1);                  \
                break;                                                                                                \
            }                                                                                                         \
            if (!c->counters_zeroed || warm_fam > 0) HIPCHK(c, hipMemsetAsync(c->wspc.p, 0, 8 * sizeof(int), s));    /* (k_prep_joint cleared them for the step's first family) */ \
            warm_fam++;                                                                                               \
            WarmSpill A0 = spill(c->wspA, 0), B1 = spill(c->wspB, 1), A2 = spill(c->wspA, 2);                         \
            const unsigned gw = (unsigned)((nit + 63) / 64);                                                          \
            dim3 g2(std::max(64u, gw / 3)), g3(std::max(32u, gw / 8)), g4(std::max(16u, gw / 24));                    \
            if (sphere) {                                                                                             \
                RFS_LAUNCH_WARM1(FAM, true, true, grid, QQ, MDLC, sp_, none, A0, wb1, 0);                             \
                if (wb2 > 0) RFS_LAUNCH_WARM1(FAM, true, false, g2, QQ, MDLC, sp_, A0, B1, wb2, 1);                    \
                if (wb2 > 0 && wb3 > 0) RFS_LAUNCH_WARM1(FAM, true, false, g3, QQ, MDLC, sp_, B1, A2, wb3, 2);         \
                if (coop_last) RFS_LAUNCH_WARMC(FAM, QQ, MDLC, (wb2 > 0 ? (wb3 > 0 ? A2 : B1) : A0));                             \
                else RFS_LAUNCH_WARM1(FAM, true, false, (wb2 > 0 ? (wb3 > 0 ? g4 : g3) : g2), QQ, MDLC, sp_, (wb2 > 0 ? (wb3 > 0 ? A2 : B1) : A0), none, NOLIM, 3); \
            } else {                                                                                                  \
                RFS_LAUNCH_WARM1(FAM, false, true, grid, QQ, MDLC, sp_, none, A0, wb1, 0);                            \
                if (wb2 > 0) RFS_LAUNCH_WARM1(FAM, false, false, g2, QQ, MDLC, sp_, A0, B1, wb2, 1);                   \
                if (wb2 > 0 && wb3 > 0) RFS_LAUNCH_WARM1(FAM, false, false, g3, QQ, MDLC, sp_, B1, A2, wb3, 2);        \
                if (coop_last) RFS_LAUNCH_WARMC(FAM, QQ, MDLC, (wb2 > 0 ? (wb3 > 0 ? A2 : B1) : A0));                             \
                else RFS_LAUNCH_WARM1(FAM, false, false, (wb2 > 0 ? (wb3 > 0 ? g4 : g3) : g2), QQ, MDLC, sp_, (wb2 > 0 ? (wb3 > 0 ? A2 : B1) : A0), none, NOLIM, 3); \
            }                                                                                                         \
        } while (0)
        int warm_fam = 0;
        if (Q.nper_total > 0) RFS_LAUNCH_WARM(SwdRayFamily, Q, c->mdlc.as<double>(), c->sphR.as<double>());
        if (P.QL.nper_total > 0) RFS_LAUNCH_WARM(SwdLoveFamily, P.QL, c->mdlcL.as<double>(), c->sphL.as<double>());
#undef RFS_LAUNCH_WARM1
#undef RFS_LAUNCH_WARMC
#undef RFS_LAUNCH_WARM
        HIPCHK(c, hipGetLastError());
        if (sb) {
            const int cap = nchain;
            ENSURE(c, c->cold_roots, (size_t)cap * np_cold * COLD_NR * 2 * sizeof(double));
            ENSURE(c, c->cold_s0, (size_t)cap * np_cold * sizeof(int));
            {   // (the counts start from zero and k_swd_cold_pick leaves zeros behind)
                const size_t before = c->cold_nroot.cap;
                ENSURE(c, c->cold_nroot, (size_t)cap * np_cold * sizeof(int));
                if (c->cold_nroot.cap != before) HIPCHK(c, hipMemsetAsync(c->cold_nroot.p, 0, c->cold_nroot.cap, s));
            }
            if (!c->cold_ticket.p) { ENSURE(c, c->cold_ticket, 4 * sizeof(int)); HIPCHK(c, hipMemsetAsync(c->cold_ticket.p, 0, 4 * sizeof(int), s)); }
            auto gscan = [&](const SwdSeqs& QQ) { return (unsigned)std::min<size_t>((size_t)cap * QQ.nper_total * COLD_TP, (size_t)8192); };
            auto gpick = [&](const SwdSeqs& QQ) { return (unsigned)std::min<size_t>((size_t)cap * QQ.nseq, (size_t)2048); };
            auto lpick = [&](const SwdSeqs& QQ) { int m = 1; for (int q = 0; q < QQ.nseq; q++) m = std::max(m, QQ.s[q].nper);
                                                  return (size_t)m * (COLD_NR * 2 * sizeof(double) + 3 * sizeof(int)) + 16; };
            const int nblk = (Q.nper_total > 0 ? (int)gpick(Q) : 0) + (P.QL.nper_total > 0 ? (int)gpick(P.QL) : 0);
            SwdCold C{c->cold_roots.as<double>(), c->cold_nroot.as<int>(), c->cold_s0.as<int>(), cap, c->cold_ticket.as<int>(), nblk};
#define RFS_LAUNCH_COLD(FAM, QQ, MDL, MDLC, SFL)                                                                       \
            do {                                                                                                       \
                hipLaunchKernelGGL((k_swd_cold_scan<FAM>), dim3(gscan(QQ)), dim3(64), 0, s, nchain, n, QQ, MDL, MDLC, W, C);   \
                hipLaunchKernelGGL((k_swd_cold_pick<FAM>), dim3(gpick(QQ)), dim3(64), lpick(QQ), s, nchain, n, QQ, MDL, MDLC, \
                                   c->croot.as<double>(), SFL, W, C);                                                   \
            } while (0)
            if (Q.nper_total > 0) RFS_LAUNCH_COLD(SwdRayFamily, Q, mdlR, c->mdlc.as<double>(), c->sflag.as<int>());
            if (P.QL.nper_total > 0) RFS_LAUNCH_COLD(SwdLoveFamily, P.QL, c->mdlL.as<float>(), c->mdlcL.as<double>(), sflagL);
#undef RFS_LAUNCH_COLD
            HIPCHK(c, hipGetLastError());
        }
        // The hand-back lists are nearly always empty, and when one is not, the full search of even ONE chain takes ~3 ms
        // (about a thousand dependent secular evaluations): it runs on a side stream -- for the chains k_swd_warm itself
        // declines (moves too large for a first-order model: the bulk) from here on, beside the branch test; for the chains
        // the branch test declines behind it -- beside the eigenfunction pass of all chains; only the listed chains'
        // eigenfunctions are redone afterwards (below).
        hipStream_t sf = (!sb && kernels && c->stream_l && s != c->stream_l) ? c->stream_l : s;
        const bool async = c->flow_cur && c->flow_async && kernels && sf != s;
        if (async) {
            // background form: ONE list for the three kinds of hand-back and ONE search behind the reference-root stage.  (On the
            // same side stream as the foreground form: more streams than hardware queues -- four by default -- would put the
            // search into the queue of the main or the surface-wave stream and serialise it with the step after all: measured.)
            W.list2 = W.list3 = W.list; W.count2 = W.count3 = W.count;
        }
        warm_side = sf != s ? sf : nullptr;
        // flow entries with "flow_async_handback": the handed-back chains sit this step out (k_flow_post) and their search
        // runs on the side stream beside the NEXT step, whose eigenfunction pass waits for it; nobody waits here
        bg_record = async;
        c->last_async = bg_record;
        // the first list's length of an earlier step, whenever its copy has arrived (never waited for)
        const int est = std::max(c->warm_est, 64);
        auto launch_fallback = [&](const int* list, const int* count, int estc) -> int {
            const int gl = std::min((nchain + 63) / 64, std::max(8, (estc + 63) / 64));
            bool rdone = false;
            if (Q.nseq > 0 && estc * Q.nseq > SWD_LAT_MAX_ITEMS) {
                // many chains handed back (large steps): the cooperative blocks of the full search, over the list
                const CoopPlan cp = coop_plan(c, Q, nchain, n);
                if (cp.ok) {
                    dim3 grid(cp.blocks);
                    size_t lds2 = cp.lds;
#define RFS_LAUNCH_COOPL(NCH)                                                                                  \
                    do {                                                                                       \
                        HIPCHK(c, hipFuncSetAttribute((const void*)k_swd_roots_coop<SwdRayFamily, NCH>,        \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));  \
                        hipLaunchKernelGGL((k_swd_roots_coop<SwdRayFamily, NCH>), grid, dim3(512), lds2, sf, nchain, n, Q, \
                                           mdlR, c->mdlc.as<double>(), c->croot.as<double>(), c->sflag.as<int>(), list, count); \
                    } while (0)
                    if (cp.nch <= 5) RFS_LAUNCH_COOPL(5);
                    else if (cp.nch <= 8) RFS_LAUNCH_COOPL(8);
                    else RFS_LAUNCH_COOPL(16);
#undef RFS_LAUNCH_COOPL
                    rdone = true;
                }
            }
            if (!rdone && Q.nseq > 0 && (n < 3 || launch_roots_split<SwdRayFamily>(c, sf, nchain, n, Q, mdlR, c->mdlc.as<double>(),
                                                                                   c->sflag.as<int>(), 0, list, count, estc)))
                hipLaunchKernelGGL((k_swd_roots<false, false>), dim3(gl * Q.nseq), dim3(64), 0, sf, nchain, n, Q, mdlR, c->croot.as<double>(),
                                   c->sflag.as<int>(), list, count, (double*)nullptr, 1);
            const bool ldone = P.QL.nseq > 0 && estc * P.QL.nseq > SWD_LAT_MAX_ITEMS &&
                               !launch_love_coop(c, sf, nchain, n, P.QL, c->mdlL.as<float>(), c->mdlcL.as<double>(), sflagL, list, count, estc);
            if (!ldone && P.QL.nseq > 0 && (n < 3 || launch_roots_split<SwdLoveFamily>(c, sf, nchain, n, P.QL, c->mdlL.as<float>(),
                                                                                       c->mdlcL.as<double>(), sflagL, 0, list, count, estc)))
                hipLaunchKernelGGL((k_swd_roots<true, false>), dim3(gl * P.QL.nseq), dim3(64), 0, sf, nchain, n, P.QL, c->mdlL.as<float>(),
                                   c->croot.as<double>(), sflagL, list, count, (double*)nullptr, 1);
            HIPCHK(c, hipGetLastError());
            return RFS_OK;
        };
        if (sf != s && !async) { HIPCHK(c, hipEventRecord(c->ev_w[0], s)); HIPCHK(c, hipStreamWaitEvent(sf, c->ev_w[0], 0)); }
        if (sf != s && !async) TRY(launch_fallback(W.list, W.count, est));      // (one stream only: both lists after the branch test)
        if (c->h_wcount && sf != s && !async) {
            if (*c->h_wcount >= 0) c->warm_est = *c->h_wcount;
            HIPCHK(c, hipMemcpyAsync(c->h_wcount, W.count, sizeof(int), hipMemcpyDeviceToHost, sf));
        }
        // ... and is the continued root still the one the reference's scan would stop at?  (one evaluation per item)
        if (Q.nper_total > 0)
            hipLaunchKernelGGL((k_swd_warm_check<SwdRayFamily>), dim3((unsigned)(((size_t)Q.nper_total * nchain + 63) / 64)), dim3(64), 0, s,
                               nchain, n, Q, mdlR, c->mdlc.as<double>(), c->croot.as<double>(), W);
        if (P.QL.nper_total > 0)
            hipLaunchKernelGGL((k_swd_warm_check<SwdLoveFamily>), dim3((unsigned)(((size_t)P.QL.nper_total * nchain + 63) / 64)), dim3(64), 0, s,
                               nchain, n, P.QL, c->mdlL.as<float>(), c->mdlcL.as<double>(), c->croot.as<double>(), W);
        // ... sequences with anomalous dispersion: the reference's scan grid itself -- 64 lanes per item for the first period
        // of a sequence (a scan of ~100 cells), 16 for the others (grids: sized for a share of the chains, the number of
        // irregular ones is only known on the device -- the blocks stride)
        // (background form with a stream to spare: the walk runs BESIDE the reference-root stage -- that stage does not need its
        // verdicts, a chain the walk declines afterwards just sits the step out like any other -- and the search of the
        // handed-back chains and the end of the step wait for both)
        hipStream_t sw = (async && c->stream_w) ? c->stream_w : s;      // (without the reference-root stage: beside the eigenfunction pass)
        if (sw != s) { HIPCHK(c, hipEventRecord(c->ev_wk[0], s)); HIPCHK(c, hipStreamWaitEvent(sw, c->ev_wk[0], 0)); }
        {
            const int gw = std::max(256, std::min(4096, (int)(((size_t)std::max(Q.nper_total, P.QL.nper_total) * nchain / 4 + 15) / 16)));
            const int g1 = std::max(64, std::min(2048, nchain / 4));
            const int ipb = sb ? 8 : 64;                     // items per block and trip (k_swd_warm_walk_dense)
            const int gd = sb ? std::max(1, std::min(4096, (int)(((size_t)std::max(Q.nper_total, P.QL.nper_total) * nchain + ipb - 1) / ipb)))
                              : std::max(256, std::min(4096, (int)(((size_t)std::max(Q.nper_total, P.QL.nper_total) * nchain / 2 + 63) / 64)));
            if (Q.nper_total > 0) {
                hipLaunchKernelGGL((k_swd_warm_walk<SwdRayFamily, true>), dim3(g1), dim3(64), 0, sw, nchain, n, Q, mdlR, c->mdlc.as<double>(),
                                   c->croot.as<double>(), W);
                if (c->walk_dense) hipLaunchKernelGGL((k_swd_warm_walk_dense<SwdRayFamily>), dim3(gd), dim3(64), 0, sw, nchain, n, Q, mdlR,
                                                      c->mdlc.as<double>(), c->croot.as<double>(), W, ipb);
                else hipLaunchKernelGGL((k_swd_warm_walk<SwdRayFamily, false>), dim3(gw), dim3(64), 0, sw, nchain, n, Q, mdlR, c->mdlc.as<double>(),
                                        c->croot.as<double>(), W);
            }
            if (P.QL.nper_total > 0) {
                hipLaunchKernelGGL((k_swd_warm_walk<SwdLoveFamily, true>), dim3(g1), dim3(64), 0, sw, nchain, n, P.QL, c->mdlL.as<float>(),
                                   c->mdlcL.as<double>(), c->croot.as<double>(), W);
                if (c->walk_dense) hipLaunchKernelGGL((k_swd_warm_walk_dense<SwdLoveFamily>), dim3(gd), dim3(64), 0, sw, nchain, n, P.QL,
                                                      c->mdlL.as<float>(), c->mdlcL.as<double>(), c->croot.as<double>(), W, ipb);
                else hipLaunchKernelGGL((k_swd_warm_walk<SwdLoveFamily, false>), dim3(gw), dim3(64), 0, sw, nchain, n, P.QL, c->mdlL.as<float>(),
                                        c->mdlcL.as<double>(), c->croot.as<double>(), W);
            }
        }
        if (sw != s) HIPCHK(c, hipEventRecord(c->ev_wk[1], sw));
        HIPCHK(c, hipGetLastError());
        // the chains the branch test handed back (and, on one stream, those of the first list)
        if (async) {}
        else if (sf != s) { HIPCHK(c, hipEventRecord(c->ev_w[1], s)); HIPCHK(c, hipStreamWaitEvent(sf, c->ev_w[1], 0)); }
        else if (!(sb && c->warm_exact)) TRY(launch_fallback(W.list, W.count, est));      // (sb: one search, behind the reference-root stage)
        if (!async && !sb) TRY(launch_fallback(W.list2, W.count2, 64));
        delete tw; tw = nullptr;
        if (c->warm_exact) {
            KTimer tx(c, RFS_K_SWD_EXACT, s);
            // ... and from the continued roots to the reference's own: its refinement (nevill) inside its scan cell, groups of
            // periods per lane (k_swd_exact); what a lane declines goes to the full search like the branch test's chains
            int G = std::max(1, c->exact_group);
            // (small batches with the second try below: the stage is as long as a group's periods + run-up, one after the other --
            // groups of 2 instead of 4: ONE configs[0] chain 0.76 -> 0.68 ms per evaluation)
            if (sb && c->exact_coop != 0 && n >= 3 && n - 1 <= 64) G = std::max(1, std::min(G, c->exact_group_small));
            const int ru = std::max(0, c->exact_runup);
            auto ngroups = [&](const SwdSeqs& QQ, int g_) { int g = 0; for (int q = 0; q < QQ.nseq; q++) g += (QQ.s[q].nper + g_ - 1) / g_; return g; };
            // Small batches (round 6): a lane per group leaves the chip empty and the stage as long as ever -- its length is one
            // lane's ~40 dependent evaluations, whatever the number of lanes.  Up to EXACT_COOP_MAX groups take 16 lanes each
            // (k_swd_exact_coop: a third of the time per evaluation, bit-identical).
            const int np_max = std::max(Q.nper_total, P.QL.nper_total);
            bool coop = c->exact_coop != 0 && n >= 3 && n - 1 <= 64 &&
                        (c->exact_coop > 1 || (size_t)ngroups(np_max == Q.nper_total ? Q : P.QL, G) * nchain <= (size_t)EXACT_COOP_MAX);
            // (shorter groups: every group has its own run-up, and a run-up that does not contract used to hand the chain to the
            // sequential search -- ONE chain of configs[0]: 27 -> 96 of 299 evaluations; with a second try behind eight run-up
            // periods none does)
            if (!coop) G = std::max(2, G);
            // big batches in rounds ("swd_exact_budget"): a budget of evaluations per lane, the unfinished groups continued with 16
            // lanes each.  Lists of a quarter of the groups; one that overflows is not an error (those groups finish in place).
            const int budget = (!coop && c->exact_budget > 0 && n - 1 <= 64) ? c->exact_budget : 0;
            // (the second launch on the walk stream, beside the eigenfunction pass: the background form of the flow entries with a
            // stream to spare -- and no second try of groups (ru2), which would want this launch's results first)
            const bool x2 = budget > 0 && sw != s && kernels && c->exact_overlap && c->exact_redo_runup <= ru && c->ev_x1 != nullptr &&
                            !c->swd_water_cur;
            c->xg[0].on = c->xg[1].on = false;
            const size_t xcap = budget ? std::max<size_t>(4096, (size_t)ngroups(np_max == Q.nper_total ? Q : P.QL, G) * nchain / 4) : 1;
            if (budget) {
                ENSURE(c, c->xsp, xcap * (EXACT_SPILL_ND + 1) * sizeof(double));
            }
            // groups whose run-up did not contract: done again with a longer one ("swd_exact_redo_runup") instead of handing
            // their chain to the sequential search
            // (-1 = automatic: 4 run-up periods -- 8 in the small batches of "swd_cold_scan", whose groups are shorter -- for the second
            // try of the small batches' 16-lane form -- ONE configs[0] chain hands
            // 15 % of its evaluations back for this cause alone -- and none for big batches, where the cause is 0.1 chains per
            // step and the list's launch would sit on the step's critical chain)
            const int ru2_opt = c->exact_redo_runup < 0 ? (coop ? (sb ? 8 : 4) : 0) : c->exact_redo_runup;
            const int ru2 = ru2_opt > ru ? ru2_opt : 0;
            const size_t rcap = ru2 ? std::max<size_t>(1024, (size_t)ngroups(np_max == Q.nper_total ? Q : P.QL, G) * nchain / 8) : 1;
            ENSURE(c, c->xspc, 4 * sizeof(int));
            if (ru2) ENSURE(c, c->xredo, 2 * rcap * sizeof(int));
            if (!c->counters_zeroed) HIPCHK(c, hipMemsetAsync(c->xspc.p, 0, 4 * sizeof(int), s));      // (k_prep_joint cleared them)
#define RFS_LAUNCH_EXACT(FAM, QQ, MDL, MDLC, CI)                                                                        \
            do {                                                                                                       \
                const int ng = ngroups(QQ, G);                                                                          \
                const size_t ldsb = (size_t)4 * ((size_t)(n - 1) * FAM::NENT + FAM::NV) * sizeof(double);               \
                const ExactSpill none{nullptr, nullptr, c->xspc.as<int>() + (CI), 0};                                   \
                const ExactRedo noredo{nullptr, nullptr, 0};                                                            \
                const ExactRedo rd = ru2 ? ExactRedo{c->xredo.as<int>() + (size_t)(CI) * rcap, c->xspc.as<int>() + 2 + (CI), (int)rcap} : noredo; \
                if (coop) {                                                                                             \
                    const unsigned gx = (unsigned)std::min<size_t>(((size_t)ng * nchain + 3) / 4, (size_t)4096);        \
                    hipLaunchKernelGGL((k_swd_exact_coop<FAM>), dim3(std::max(1u, gx)), dim3(64), ldsb, s, nchain, n, QQ, G, ru, ng, \
                                       c->exact_origin_tol, MDL, MDLC, c->croot.as<double>(), W, (const int*)nullptr, (const int*)nullptr, none, rd); \
                } else if (budget) {                                                                                    \
                    double* xd = c->xsp.as<double>();                                                                   \
                    ExactSpill sp{xd, (unsigned long long*)(xd + (size_t)EXACT_SPILL_ND * xcap), c->xspc.as<int>() + (CI), (int)xcap}; \
                    hipLaunchKernelGGL((k_swd_exact<FAM>), dim3((unsigned)(((size_t)ng * nchain + 63) / 64)), dim3(64), 0, s, \
                                       nchain, n, QQ, G, ru, ng, c->exact_origin_tol, MDL, MDLC, c->croot.as<double>(), W, sp, budget, rd); \
                    const unsigned gx = (unsigned)std::min<size_t>((xcap + 3) / 4, (size_t)4096);                       \
                    /* the second launch beside the eigenfunction pass of all items (x2: the walk stream, idle by now): the   \
                       groups it finishes get their eigenfunctions afterwards (k_swd_eigen_groups, below) */                \
                    hipStream_t sx = x2 ? sw : s;                                                                       \
                    if (x2) { HIPCHK(c, hipEventRecord(c->ev_x1, s)); HIPCHK(c, hipStreamWaitEvent(sx, c->ev_x1, 0)); }    \
                    hipLaunchKernelGGL((k_swd_exact_coop<FAM>), dim3(gx), dim3(64), ldsb, sx, nchain, n, QQ, G, ru, ng,   \
                                       c->exact_origin_tol, MDL, MDLC, c->croot.as<double>(), W, (const int*)nullptr, (const int*)nullptr, sp, rd); \
                    if (x2) { c->xg[CI] = rfs_ctx::XGroups{sp.item, sp.count, (int)xcap, G, true}; }                                \
                } else {                                                                                                \
                    hipLaunchKernelGGL((k_swd_exact<FAM>), dim3((unsigned)(((size_t)ng * nchain + 63) / 64)), dim3(64), 0, s, \
                                       nchain, n, QQ, G, ru, ng, c->exact_origin_tol, MDL, MDLC, c->croot.as<double>(), W, none, 0x7fffffff, rd); \
                }                                                                                                       \
                if (ru2) {                                                                                              \
                    const unsigned gr = (unsigned)std::min<size_t>((rcap + 3) / 4, (size_t)1024);                       \
                    hipLaunchKernelGGL((k_swd_exact_coop<FAM>), dim3(gr), dim3(64), ldsb, s, nchain, n, QQ, G, ru2, ng,   \
                                       c->exact_origin_tol, MDL, MDLC, c->croot.as<double>(), W, (const int*)rd.list, (const int*)rd.count, none, noredo); \
                }                                                                                                       \
            } while (0)
            if (Q.nper_total > 0) RFS_LAUNCH_EXACT(SwdRayFamily, Q, mdlR, c->mdlc.as<double>(), 0);
            if (P.QL.nper_total > 0) RFS_LAUNCH_EXACT(SwdLoveFamily, P.QL, c->mdlL.as<float>(), c->mdlcL.as<double>(), 1);
#undef RFS_LAUNCH_EXACT
            if (x2) HIPCHK(c, hipEventRecord(c->ev_wk[1], sw));      // (the walk stream's work now ends with the stage's second launch)
            HIPCHK(c, hipGetLastError());
            if (sf != s) { HIPCHK(c, hipEventRecord(c->ev_w[4], s)); HIPCHK(c, hipStreamWaitEvent(sf, c->ev_w[4], 0)); }
            if (!async) TRY(launch_fallback(W.list3, W.count3, sb ? est : 64));
        }
        walk_join = sw != s;
        if (async) {
            if (!c->warm_exact) { HIPCHK(c, hipEventRecord(c->ev_w[4], s)); HIPCHK(c, hipStreamWaitEvent(sf, c->ev_w[4], 0)); }
            if (walk_join) HIPCHK(c, hipStreamWaitEvent(sf, c->ev_wk[1], 0));
            TRY(launch_fallback(W.list, W.count, est));
            if (c->h_wcount) {
                if (*c->h_wcount >= 0) c->warm_est = *c->h_wcount;
                HIPCHK(c, hipMemcpyAsync(c->h_wcount, W.count, sizeof(int), hipMemcpyDeviceToHost, sf));
            }
        }
        roots = false;
    }
    // the two families' searches are independent: outside the CU-partitioned step the Love one runs on its own stream beside
    // the Rayleigh one (a fifth active stream inside the partitioned step would share a hardware queue)
    // -- inside it the Love search goes to the RF half's stream, ahead of the RF sweeps: the Rayleigh search keeps its half
    if (roots && c->swd_water_cur && c->swd_mode_cur == 0) {
        // models with a water layer on top (vs(1) = 0, surfdisp96.f:138-139): the lane-per-item kernel, whose secular
        // functions carry the reference's water branch (:870-886; Love: the layers below the water, :750)
        KTimer t(c, RFS_K_SWD_ROOTS, s);
        if (Q.nseq > 0)
            hipLaunchKernelGGL((k_swd_roots<false, false>), dim3((Q.nseq * nchain + 63) / 64), dim3(64), 0, s, nchain, n, Q, mdlR,
                               c->croot.as<double>(), c->sflag.as<int>(), (const int*)nullptr, (const int*)nullptr, (double*)nullptr, 1);
        if (P.QL.nseq > 0)
            hipLaunchKernelGGL((k_swd_roots<true, false>), dim3((P.QL.nseq * nchain + 63) / 64), dim3(64), 0, s, nchain, n, P.QL,
                               c->mdlL.as<float>(), c->croot.as<double>(), sflagL, (const int*)nullptr, (const int*)nullptr, (double*)nullptr, 1);
        HIPCHK(c, hipGetLastError());
        roots = false;
    }
    if (roots && c->swd_mode_cur > 0) {
        // higher modes (libsurf's `mode` argument): the mode loop runs inside the lane-per-item kernel's state machine
        // (latency form only: B1 calls and plugins with SurfWD(mode = ...); not a bench configuration)
        KTimer t(c, RFS_K_SWD_ROOTS, s);
        ENSURE(c, c->craw, (size_t)P.nitems * nchain * sizeof(double));
        const int nmode = c->swd_mode_cur + 1;
        if (Q.nseq > 0)
            hipLaunchKernelGGL((k_swd_roots<false, true>), dim3((Q.nseq * nchain + 63) / 64), dim3(64), 0, s, nchain, n, Q, mdlR,
                               c->croot.as<double>(), c->sflag.as<int>(), (const int*)nullptr, (const int*)nullptr,
                               c->craw.as<double>(), nmode);
        if (P.QL.nseq > 0)
            hipLaunchKernelGGL((k_swd_roots<true, true>), dim3((P.QL.nseq * nchain + 63) / 64), dim3(64), 0, s, nchain, n, P.QL,
                               c->mdlL.as<float>(), c->croot.as<double>(), sflagL, (const int*)nullptr, (const int*)nullptr,
                               c->craw.as<double>(), nmode);
        HIPCHK(c, hipGetLastError());
        roots = false;
    }
    const bool love_aside = roots && P.QL.nseq > 0 && Q.nseq > 0 && c->stream_l && s != c->stream3;
    if (roots && P.QL.nseq > 0) {       // Love: 2-vector recurrence, same lanes-per-item search as the small Rayleigh batches
        hipStream_t sl = !love_aside ? s : (s == c->stream2m ? c->stream3 : c->stream_l);
        if (love_aside) { HIPCHK(c, hipEventRecord(c->ev_lf, s)); HIPCHK(c, hipStreamWaitEvent(sl, c->ev_lf, 0)); }
        {
            KTimer t(c, RFS_K_SWD_ROOTS, sl);
            int nitem = P.QL.nseq * nchain;
            if (launch_love_coop(c, sl, nchain, n, P.QL, c->mdlL.as<float>(), c->mdlcL.as<double>(), sflagL) &&
                (n < 3 || launch_roots_split<SwdLoveFamily>(c, sl, nchain, n, P.QL, c->mdlL.as<float>(), c->mdlcL.as<double>(),
                                                            sflagL, c->swd_lanes)))
                hipLaunchKernelGGL((k_swd_roots<true, false>), dim3((nitem + 63) / 64), dim3(64), 0, sl, nchain, n, P.QL,
                                   c->mdlL.as<float>(), c->croot.as<double>(), sflagL, (const int*)nullptr, (const int*)nullptr,
                                   (double*)nullptr, 1);
            HIPCHK(c, hipGetLastError());
        }
        if (love_aside) HIPCHK(c, hipEventRecord(c->ev_lj, sl));
    }
    if (roots && Q.nseq > 0) {
        KTimer t(c, RFS_K_SWD_ROOTS, s);
        int nitem = Q.nseq * nchain;
        CoopPlan cp = coop_plan(c, Q, nchain, n);
        if (cp.ok) {
            // cooperative producer/consumer blocks (64 items each): least total work, shortest serial path
            dim3 grid(cp.blocks);
            size_t lds2 = cp.lds;
#define RFS_LAUNCH_COOP(NCH)                                                                                   \
            do {                                                                                               \
                HIPCHK(c, hipFuncSetAttribute((const void*)k_swd_roots_coop<SwdRayFamily, NCH>,                              \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));          \
                hipLaunchKernelGGL((k_swd_roots_coop<SwdRayFamily, NCH>), grid, dim3(512), lds2, s, nchain, n, Q,               \
                                   mdlR, c->mdlc.as<double>(), c->croot.as<double>(),                           \
                                   c->sflag.as<int>(), (const int*)nullptr, (const int*)nullptr);               \
            } while (0)
            if (cp.nch <= 5) RFS_LAUNCH_COOP(5);
            else if (cp.nch <= 8) RFS_LAUNCH_COOP(8);
            else RFS_LAUNCH_COOP(16);
#undef RFS_LAUNCH_COOP
        } else if (n < 3 || launch_roots_split<SwdRayFamily>(c, s, nchain, n, Q, mdlR, c->mdlc.as<double>(), c->sflag.as<int>(),
                                                             c->swd_lanes)) {
            hipLaunchKernelGGL((k_swd_roots<false, false>), dim3((nitem + 63) / 64), dim3(64), 0, s, nchain, n, Q,
                               mdlR, c->croot.as<double>(), c->sflag.as<int>(), (const int*)nullptr, (const int*)nullptr,
                               (double*)nullptr, 1);
        }
        HIPCHK(c, hipGetLastError());
    }
    // (in the partitioned step the caller's stream joins the RF half's stream, which carries the Love search, anyway)
    if (love_aside && s != c->stream2m) HIPCHK(c, hipStreamWaitEvent(s, c->ev_lj, 0));
    if (kernels) {
        size_t ntot = (size_t)P.nitems * nchain;
        ENSURE(c, c->cds, ntot * 6 * n * sizeof(double));
        ENSURE(c, c->krn, ntot * 4 * n * sizeof(double));
        ENSURE(c, c->ugr, 3 * ntot * sizeof(double));      // U and the two kernel scale slots of every item (swd_krn)
        const double* crT_arg = c->krn_ruled ? c->crT.as<double>() : (const double*)nullptr;       // chain-ruled storage (swd_eigen_lane)
        {   // (timed group: the launches of this stream only -- the wait for the side stream below is not kernel time)
        KTimer t(c, eigen_mode == 1 ? -1 : RFS_K_SWD_EIGEN, s);     // the early launch hides behind the search: not timed
#define RFS_LAUNCH_EIGEN2(LOVE, SPH, WAT, QQ, SPHP, SFL, EL1, EARLY, EDONE)                                          \
        hipLaunchKernelGGL((k_swd_eigen<LOVE, SPH, false, WAT>), dim3((unsigned)(((size_t)(EL1) * nchain + 63) / 64)), \
                           dim3(64), 0, s, nchain, n, QQ, ntot, c->mdl.as<float>(), SPHP, c->croot.as<double>(),     \
                           SFL, c->cds.as<double>(), c->krn.as<double>(), c->ugr.as<double>(), 0, (int)(EL1), EARLY, EDONE, \
                           (const int*)nullptr, (const int*)nullptr, crT_arg)
#define RFS_LAUNCH_EIGEN(LOVE, SPH, QQ, SPHP, SFL, EL1, EARLY, EDONE)                                                \
        do { if (c->swd_water_cur) RFS_LAUNCH_EIGEN2(LOVE, SPH, true, QQ, SPHP, SFL, EL1, EARLY, EDONE);              \
             else RFS_LAUNCH_EIGEN2(LOVE, SPH, false, QQ, SPHP, SFL, EL1, EARLY, EDONE); } while (0)
        int* ed = (eigen_mode == 1 || eigen_mode == 2) ? c->edone.as<int>() : nullptr;
        if (P.QR.nper_total > 0 && eigen_mode != 4) {
            const int el1 = eigen_mode == 1 ? early_items : P.QR.nper_total;
            const int early = eigen_mode == 1;
            if (sphere) RFS_LAUNCH_EIGEN(false, true, P.QR, c->sphR.as<double>(), c->sflag.as<int>(), el1, early, ed);
            else RFS_LAUNCH_EIGEN(false, false, P.QR, nullptr, c->sflag.as<int>(), el1, early, ed);
        }
        if (P.QL.nper_total > 0 && eigen_mode != 1 && eigen_mode != 3) {
            if (sphere) RFS_LAUNCH_EIGEN(true, true, P.QL, c->sphL.as<double>(), sflagL, P.QL.nper_total, 0, (int*)nullptr);
            else RFS_LAUNCH_EIGEN(true, false, P.QL, nullptr, sflagL, P.QL.nper_total, 0, (int*)nullptr);
        }
#undef RFS_LAUNCH_EIGEN
#undef RFS_LAUNCH_EIGEN2
        HIPCHK(c, hipGetLastError());
        }
        if (walk_join) HIPCHK(c, hipStreamWaitEvent(s, c->ev_wk[1], 0));      // (the caller joins this stream: k_flow_post needs the walk's verdicts)
        for (int fi = 0; fi < 2; fi++) {
            // the groups the reference-root stage's second launch finished beside the pass above: their periods again
            if (!c->xg[fi].on) continue;
            c->xg[fi].on = false;
            const rfs_ctx::XGroups& xg = c->xg[fi];
            const unsigned gq = (unsigned)std::min<size_t>(((size_t)xg.cap * xg.G + 63) / 64, (size_t)2048);
            const SwdSeqs& QQ = fi == 0 ? P.QR : P.QL;
            const int* sfl = fi == 0 ? c->sflag.as<int>() : sflagL;
            const double* sp_ = sphere ? (fi == 0 ? c->sphR.as<double>() : c->sphL.as<double>()) : (const double*)nullptr;
#define RFS_LAUNCH_EIGEN_G(LOVE, SPH)                                                                                 \
            hipLaunchKernelGGL((k_swd_eigen_groups<LOVE, SPH>), dim3(gq), dim3(64), 0, s, nchain, n, QQ, ntot, c->mdl.as<float>(), sp_, \
                               c->croot.as<double>(), sfl, c->cds.as<double>(), c->krn.as<double>(), c->ugr.as<double>(),      \
                               xg.item, xg.count, xg.cap, xg.G, crT_arg)
            if (fi == 0) { if (sphere) RFS_LAUNCH_EIGEN_G(false, true); else RFS_LAUNCH_EIGEN_G(false, false); }
            else { if (sphere) RFS_LAUNCH_EIGEN_G(true, true); else RFS_LAUNCH_EIGEN_G(true, false); }
#undef RFS_LAUNCH_EIGEN_G
            HIPCHK(c, hipGetLastError());
        }
        if (bg_record) {
            HIPCHK(c, hipEventRecord(c->ev_bg[c->wpar], warm_side));
            c->bg_busy[c->wpar] = true;
            warm_side = nullptr;
        }
        if (warm_side) {
            // behind the full search of the handed-back chains: their eigenfunctions again, from their new roots (the pass
            // above has read whatever roots they had; it must have finished before these results are written)
            HIPCHK(c, hipEventRecord(c->ev_w[2], s));
            HIPCHK(c, hipStreamWaitEvent(warm_side, c->ev_w[2], 0));
            const int* wn = c->wneed.as<int>() + (size_t)c->wpar * (3 * (size_t)nchain + 4);
            const size_t lo = (size_t)c->wpar * nchain;
            const int* lists[3] = {c->wlist.as<int>() + lo, c->wlist2.as<int>() + lo, c->wlist3.as<int>() + lo};
            const int* counts[3] = {wn + nchain, wn + 2 * nchain + 2, wn + 3 * nchain + 3};
#define RFS_LAUNCH_EIGEN_LIST(LOVE, SPH, QQ, SPHP, SFL, LI)                                                          \
            hipLaunchKernelGGL((k_swd_eigen<LOVE, SPH, true>), dim3((unsigned)(((size_t)(QQ).nper_total * nchain + 63) / 64)), \
                               dim3(64), 0, warm_side, nchain, n, QQ, ntot, c->mdl.as<float>(), SPHP, c->croot.as<double>(), \
                               SFL, c->cds.as<double>(), c->krn.as<double>(), c->ugr.as<double>(), 0, (QQ).nper_total, 0, \
                               (int*)nullptr, lists[LI], counts[LI], crT_arg)
            for (int li = 0; li < (c->warm_exact ? 3 : 2); li++) {
                if (P.QR.nper_total > 0) {
                    if (sphere) RFS_LAUNCH_EIGEN_LIST(false, true, P.QR, c->sphR.as<double>(), c->sflag.as<int>(), li);
                    else RFS_LAUNCH_EIGEN_LIST(false, false, P.QR, (const double*)nullptr, c->sflag.as<int>(), li);
                }
                if (P.QL.nper_total > 0) {
                    if (sphere) RFS_LAUNCH_EIGEN_LIST(true, true, P.QL, c->sphL.as<double>(), sflagL, li);
                    else RFS_LAUNCH_EIGEN_LIST(true, false, P.QL, (const double*)nullptr, sflagL, li);
                }
            }
#undef RFS_LAUNCH_EIGEN_LIST
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipEventRecord(c->ev_w[3], warm_side));
            HIPCHK(c, hipStreamWaitEvent(s, c->ev_w[3], 0));
        }
    }
    return RFS_OK;
}

int check_batch(rfs_ctx* c, int nchain, int nlayer) {
    if (!c) return RFS_ERR_ARG;
    if (nchain < 1 || nchain > c->max_chains) return fail(c, RFS_ERR_ARG, "nchain outside [1, max_chains]");
    if (nlayer < 2 || nlayer > c->max_layers || nlayer > MAXL) return fail(c, RFS_ERR_ARG, "nlayer outside [2, min(max_layers,128)]");
    return RFS_OK;
}

int upload(rfs_ctx* c, Buf& b, const void* host, size_t bytes) {
    ENSURE(c, b, bytes);
    HIPCHK(c, hipMemcpyAsync(b.p, host, bytes, hipMemcpyHostToDevice, c->stream));
    return RFS_OK;
}

// the whole misfit+gradient evaluation on device pointers
// traj: 1 = the call continues a trajectory of the SAME nchain chains (leapfrog / flow entries): the root search may
// start from the previous evaluation (option "swd_warm_start"); 2 = the start models of a batch of trajectories
// (rfs_leapfrog_dev): always the reference-semantics search, whose roots and kernels then seed the steps; 0 = a
// plugin evaluation
int joint_eval(rfs_ctx* c, int nchain, const double* x, double* misfit, double* grad, double* dsyn, int32_t* flag,
               int traj = 0, const FlowPre* fpre = nullptr, RfReduce* defer = nullptr) {
    if (defer) defer->PG = nullptr;
    const int n = c->n;
    HIPCHK(c, hipSetDevice(c->device));
    // track: keep the model / roots / kernels of this evaluation for the next one; warm: use those of the previous one
    // (higher modes: always the reference-semantics search -- the warm start's branch test is the fundamental's)
    const bool track = c->has_swd && c->swd_mode == 0 && (c->warm_opt == 2 || (c->warm_opt == 1 && traj));
    struct ModeGuard { rfs_ctx* c; ~ModeGuard() { c->swd_mode_cur = 0; c->krn_ruled = false; } } mode_guard{c};
    c->swd_mode_cur = c->swd_mode;
    c->krn_ruled = c->has_swd;                 // k_swd_combine and the warm start read the chain-ruled storage
    if (c->has_swd) ENSURE(c, c->crT, 2 * (size_t)n * nchain * sizeof(double));
    const bool warm = track && c->warm_primed && c->warm_nchain == nchain && traj != 2;
    c->last_async = false;
    if (!fpre) c->flow_x = nullptr;            // (whatever this evaluation leaves behind is not a flow state's)
    if (track) {
        const size_t nn = (size_t)n * nchain;
        if (c->warm_nchain != nchain) c->warm_primed = false;
        // next set of hand-back flags / lists; which background searches are through by now (never waited for, except the one
        // whose set is needed again: RFS_BG_SLOTS steps old)
        c->wpar = (c->wpar + 1) % RFS_BG_SLOTS;
        if (c->bg_busy[c->wpar]) { HIPCHK(c, hipEventSynchronize(c->ev_bg[c->wpar])); c->bg_busy[c->wpar] = false; }
        c->bg_ready = 0;
        for (int i = 0; i < RFS_BG_SLOTS; i++) {
            if (c->bg_busy[i] && hipEventQuery(c->ev_bg[i]) == hipSuccess) c->bg_busy[i] = false;
            if (!c->bg_busy[i]) c->bg_ready |= 1u << i;
        }
        ENSURE(c, c->xw, 2 * nn * sizeof(double)); ENSURE(c, c->dxT, 2 * nn * sizeof(double));
        ENSURE(c, c->crT, 2 * nn * sizeof(double));
        const size_t before = c->wvalid.cap;
        ENSURE(c, c->wvalid, (size_t)nchain * sizeof(int)); ENSURE(c, c->wneed, RFS_BG_SLOTS * (3 * (size_t)nchain + 4) * sizeof(int));
        ENSURE(c, c->wlist3, RFS_BG_SLOTS * (size_t)nchain * sizeof(int));
        ENSURE(c, c->cwarm, (size_t)(4 * (c->ntw[0] + c->ntw[1] + c->ntw[2] + c->ntw[3])) * nchain * sizeof(double));
        ENSURE(c, c->wilist, (size_t)nchain * sizeof(int)); ENSURE(c, c->wlist2, RFS_BG_SLOTS * (size_t)nchain * sizeof(int));
        for (auto& e : c->ev_w) if (!e) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ENSURE(c, c->wlist, RFS_BG_SLOTS * (size_t)nchain * sizeof(int)); ENSURE(c, c->wforce, (size_t)nchain * sizeof(int));
        ENSURE(c, c->wstats, 40 * sizeof(unsigned long long));
        ENSURE(c, c->wsgn, (size_t)(4 * (c->ntw[0] + c->ntw[1] + c->ntw[2] + c->ntw[3])) * nchain);
        ENSURE(c, c->wslope, (size_t)(4 * (c->ntw[0] + c->ntw[1] + c->ntw[2] + c->ntw[3])) * nchain * sizeof(double));
        ENSURE(c, c->wbetmx, (size_t)2 * nchain * sizeof(float)); ENSURE(c, c->wsg1, (size_t)8 * nchain);
        ENSURE(c, c->wferr, (size_t)(4 * (c->ntw[0] + c->ntw[1] + c->ntw[2] + c->ntw[3])) * nchain * sizeof(double));
        if (!warm) HIPCHK(c, hipMemsetAsync(c->wferr.p, 0, c->wferr.cap, c->stream));      // (an evaluation by the full search leaves nothing to carry over)
        // slopes of the secular function belong to roots a warm search found: an evaluation by the full search leaves none
        if (!warm) HIPCHK(c, hipMemsetAsync(c->wslope.p, 0, c->wslope.cap, c->stream));
        if (c->wvalid.cap != before) {
            HIPCHK(c, hipMemsetAsync(c->wforce.p, 0, c->wforce.cap, c->stream));
            HIPCHK(c, hipMemsetAsync(c->wstats.p, 0, c->wstats.cap, c->stream));
        }
        if (!c->h_wcount) {
            HIPCHK(c, hipHostMalloc((void**)&c->h_wcount, sizeof(int), hipHostMallocDefault));
            *c->h_wcount = -1;
        }
    } else {
        c->warm_primed = false;            // an unrelated evaluation overwrites croot / krn
    }
    ENSURE(c, c->cr, (size_t)nchain * 2 * n * sizeof(double));
    if (c->has_rf) ENSURE(c, c->lc, (size_t)nchain * n * sizeof(RfLayer));
    if (c->has_swd) { ENSURE(c, c->mdl, (size_t)4 * n * nchain * sizeof(float)); ENSURE(c, c->mdlc, (size_t)6 * n * nchain * sizeof(double)); }
    const double* tw[4] = {c->d_tw[0].as<double>(), c->d_tw[1].as<double>(), c->d_tw[2].as<double>(), c->d_tw[3].as<double>()};
    if (c->has_swd) {
        size_t nn = (size_t)n * nchain;
        if (c->sphere && c->ntw[0] + c->ntw[1] > 0) ENSURE(c, c->sphR, 7 * nn * sizeof(double));
        if (c->sphere && c->ntw[2] + c->ntw[3] > 0) ENSURE(c, c->sphL, 7 * nn * sizeof(double));
    }
    SwdPlan P = make_plan(c->ntw, tw, true, c->sphere, 0, false, c->sphR.as<double>(), c->sphL.as<double>(), c->rg_alias,
                          c->lg_alias);
    const SwdSeqs& Q = P.QR;
    hipStream_t user = c->stream;
    // CU partition: the cooperative search occupies one CU per block (64 sequences); when that fits on half of the
    // chip it runs there undisturbed and the RF kernels take the other half (measured +11 % at config 2) -- as long
    // as the RF pipeline on half a chip does not take longer than the search.
    // (not on the context's own stream: there the extra masked streams were measured to share a hardware
    // queue with it and serialise -- the host-pointer entries keep the shared-CU schedule)
    const CoopPlan cp = coop_plan(c, Q, nchain, n);
    int npmax = 0;
    for (int q = 0; q < Q.nseq; q++) npmax = Q.s[q].nper > npmax ? Q.s[q].nper : npmax;
    const bool rf_time = c->has_rf && c->f.method != RFS_RF_FREQ;
    // chain tiles of the RF pipeline: the pass-A row scratch of one tile stays within rf_scratch_budget
    int rf_tile = nchain;
    // rows by peeling: a chain whose layer matrices stay close to unitary (rf_growth_exponent: teleseismic slowness, a time
    // window not much shorter than the S travel time through the stack) rebuilds its rows in pass B; the kernels decide per
    // chain, the scratch is sized for everyone and written by the others only
    const bool rf_peel = c->has_rf && !rf_time && c->rf_peel != 0;
    c->f.peel_emax = (c->rf_peel == 2) ? 1.0e300 : RF_PEEL_EMAX;
    if (c->has_rf && !rf_time) {
        const size_t per_chain = (size_t)(n - 1) * 8 * c->f.nkp * sizeof(double);
        size_t fit = per_chain ? c->rf_scratch_budget / per_chain : (size_t)nchain;
        if (fit < (size_t)nchain) rf_tile = (int)std::max<size_t>(64, fit / 64 * 64);
        rf_tile = std::min(rf_tile, RF_MAX_CHAINS_PER_LAUNCH);      // grid rows of the RF sweeps
    }
    const bool tiled = rf_tile < nchain;
    // (a warm-started search is throughput work like the RF sweeps: no partition, everything shares the chip)
    const bool part_possible = !warm && c->swd_mode == 0 && !rf_time && !tiled && !c->own_stream && c->has_rf && c->has_swd && c->cu_split && c->stream2m &&
                               c->stream3 && cp.ok && cp.blocks <= (c->ncu / 2) * cp.per_cu;
    // Early eigenfunction pass: with the partition on, the RF half of the chip finishes before the search does.  The
    // eigenfunction kernels of the first periods -- whose roots have long been final by then -- fill that gap on the RF
    // half; only the rest waits for the search (k_swd_eigen launch modes).  What is left for the mop-up should be
    // whole rounds of the chip's wave slots (an eigenfunction wavefront runs ~0.4 ms however few there are), and the
    // early part should end about when the search does.  A wrong choice costs time, never results: a wavefront whose
    // roots are not final is left to the mop-up launch.
    const bool early_possible = part_possible && P.QR.nseq == 1 && P.QL.nseq == 0 && nchain % 64 == 0;
    bool part = part_possible;
    int early_items = 0;
    StepCalib* cal = nullptr;
    bool timed = false;
    if (part_possible) {
        // One calibration per SHAPE BUCKET: chain counts within 1/8 of an octave share it (length-sorted trajectories
        // evaluate a slightly different number of chains at every leapfrog step), and at most 32 buckets are ever
        // calibrated -- a shape beyond that borrows the schedule of the nearest calibrated chain count (or runs the
        // default: partition, nothing early).
        int gran = 64;
        while (gran * 16 <= nchain) gran *= 2;
        const int bucket = (nchain + gran / 2) / gran * gran;
        const auto key = std::make_tuple(bucket, n, c->f.nft, npmax, P.nseq);
        auto it = c->calib.find(key);
        if (it == c->calib.end() && c->calib.size() >= 32) {
            const StepCalib* near = nullptr; int dist = 1 << 30;
            for (auto& kv : c->calib)
                if (std::get<1>(kv.first) == n && std::get<2>(kv.first) == c->f.nft && std::get<3>(kv.first) == npmax &&
                    std::get<4>(kv.first) == P.nseq && kv.second.stage >= CALIB_DONE &&
                    std::abs(std::get<0>(kv.first) - bucket) < dist) { near = &kv.second; dist = std::abs(std::get<0>(kv.first) - bucket); }
            if (near) { part = near->part; early_items = near->early_items; }
        } else {
            cal = &c->calib[key];
        }
    }
    if (cal) {
        // harvest the event pairs of earlier timed calls that have completed (never wait: a host that blocks here stops
        // running ahead of the device, and the launch gaps that then open up would be timed as part of the schedule)
        while (cal->stage >= 0 && cal->stage < CALIB_DONE && cal->harvested < cal->stage &&
               hipEventQuery(cal->ev[2 * cal->harvested + 1]) == hipSuccess) {
            float ms = 0.f;
            StepCand& k = cal->cand[cal->harvested / 2];
            if (hipEventElapsedTime(&ms, cal->ev[2 * cal->harvested], cal->ev[2 * cal->harvested + 1]) == hipSuccess &&
                ms > 0.f && (k.ms == 0.f || ms < k.ms))
                k.ms = ms;
            cal->harvested++;
        }
        if (cal->stage < 0) {            // first call of the shape: default schedule, untimed; build the candidate list
            cal->cand.clear();
            cal->cand.push_back(StepCand{false, 0, 0.f});
            cal->cand.push_back(StepCand{true, 0, 0.f});
            if (early_possible) {
                const int ipr = std::max(1, c->ncu * 8 / (nchain / 64));    // periods per full-chip round of wave slots
                const int step = std::max(1, std::min(ipr, npmax) / 4);
                for (int e = step; e < npmax; e += step) cal->cand.push_back(StepCand{true, e, 0.f});
            }
            for (auto e : cal->ev) hipEventDestroy(e);
            cal->ev.assign(4 * cal->cand.size(), nullptr);
            for (auto& e : cal->ev) if (hipEventCreate(&e) != hipSuccess) { cal->cand.clear(); break; }
            cal->harvested = 0;
            cal->stage = cal->cand.empty() ? CALIB_DONE : 0;
        } else if (cal->stage < CALIB_DONE) {
            const int ntimed = 2 * (int)cal->cand.size();
            if (cal->stage < ntimed) {                                   // next candidate (each one twice in a row)
                const StepCand& k = cal->cand[cal->stage / 2];
                part = k.part; early_items = k.early;
                timed = true;
            } else if (cal->harvested >= ntimed) {                       // everything measured: decide
                int best = 0;
                for (int i = 1; i < (int)cal->cand.size(); i++)
                    if (cal->cand[i].ms > 0.f && (cal->cand[best].ms == 0.f || cal->cand[i].ms < cal->cand[best].ms)) best = i;
                cal->part = cal->cand[best].part; cal->early_items = cal->cand[best].early;
                cal->stage = CALIB_DONE;
                if (getenv("RFS_DEBUG_CALIB")) {
                    fprintf(stderr, "[rfs] schedule of %d chains x %d layers, nft %d:", nchain, n, c->f.nft);
                    for (auto& k : cal->cand) fprintf(stderr, " (%s,%d) %.3f", k.part ? "split" : "shared", k.early, k.ms);
                    fprintf(stderr, " ms -> %s, early %d\n", cal->part ? "split" : "shared", cal->early_items);
                }
            }
        }
        if (cal->stage >= CALIB_DONE && !timed) { part = cal->part; early_items = cal->early_items; }
    }
    if (part_possible) {
        if (!timed && early_possible && c->early_eigen >= 0) early_items = c->early_eigen;       // explicit count (0 = off)
        if (!part || !early_possible) early_items = 0;
        early_items = std::max(0, std::min(early_items, npmax - 1));
        if (timed) HIPCHK(c, hipEventRecord(cal->ev[2 * cal->stage], user));
    }
    const bool a_eigen = part && P.QL.nseq > 0 && P.QR.nseq > 0;
    bool rf_reduced = false;
    if (early_items > 0) {
        const size_t ntot = (size_t)P.nitems * nchain;
        ENSURE(c, c->croot, ntot * sizeof(double));
        ENSURE(c, c->sflag, (size_t)8 * nchain * sizeof(int));
        ENSURE(c, c->cds, ntot * 6 * n * sizeof(double));
        ENSURE(c, c->krn, ntot * 4 * n * sizeof(double));
        ENSURE(c, c->ugr, 3 * ntot * sizeof(double));
        ENSURE(c, c->edone, (ntot / 64 + 1) * sizeof(int));
    }
    if (warm) { ENSURE(c, c->wspc, 8 * sizeof(int)); ENSURE(c, c->xspc, 4 * sizeof(int)); }
    {   // layer constants, search models -- and, for the early launch, the cleared root buffer (zero = not final) and done map
        KTimer t(c, RFS_K_PREP, c->stream);
        const size_t ntot = (size_t)P.nitems * nchain;
        int nth = nchain * n;
        hipLaunchKernelGGL(k_prep_joint, dim3((nth + 255) / 256), dim3(256), 0, c->stream, nchain, n, x,
                           (int)c->has_rf, c->f.p, c->lc.as<RfLayer>(), c->cr.as<double>(), (int)c->has_swd,
                           c->mdl.as<float>(), c->mdlc.as<double>(),
                           early_items > 0 ? c->croot.as<double>() : (double*)nullptr, early_items > 0 ? ntot : (size_t)0,
                           early_items > 0 ? c->edone.as<int>() : (warm ? c->wneed.as<int>() + (size_t)c->wpar * (3 * (size_t)nchain + 4) : (int*)nullptr),
                           early_items > 0 ? ntot / 64 + 1 : (warm ? 3 * (size_t)nchain + 4 : (size_t)0),
                           track ? c->xw.as<double>() : (double*)nullptr, c->dxT.as<double>(),
                           c->has_swd ? c->crT.as<double>() : (double*)nullptr,
                           fpre ? *fpre : FlowPre{},       // (flow entries: the step's drift rides in this kernel)
                           warm ? c->wspc.as<int>() : (int*)nullptr, warm ? c->xspc.as<int>() : (int*)nullptr);
        c->counters_zeroed = warm;
        HIPCHK(c, hipGetLastError());
        if (c->has_swd) TRY(launch_family_prep(c, c->stream, nchain, n, P, c->sphere));
    }
    // K.r of the surface-wave rows, thickness suffix sums, weighting, misfit, flags (k_swd_combine).  first: launched on the
    // surface-wave stream right behind the eigenfunction pass, BESIDE the RF sweeps -- it then writes gradient / misfit /
    // flag first and the RF reduction adds its part behind the join (k_rf_reduce, merge): the same sums, and ~0.1 ms of
    // the step's serial tail gone
    auto launch_combine = [&](hipStream_t st, int first) -> int {
        KTimer t(c, RFS_K_COMBINE, st);
        const SwdRows& R = P.R;
        const int nt = c->has_rf ? c->f.nt : 0;
        // (row cache in LDS: residual + kernel scales of every data row of the block's 32 chains, while it fits)
        const int rowc = ((size_t)(n + 3 * R.nswd) * 32 * sizeof(double) <= 56 * 1024) ? 1 : 0;
        size_t lds_c = (size_t)(n + (rowc ? 3 * R.nswd : 0)) * 32 * sizeof(double);
        // (ONE chain: its kernels, scales and roots staged in LDS -- k_swd_combine)
        const size_t lds_stage = ((size_t)R.nitems * 4 * n + (size_t)4 * R.nitems) * sizeof(double);
        const int stage = (nchain == 1 && lds_c + lds_stage <= 150 * 1024) ? 1 : 0;
        if (stage) lds_c += lds_stage;
#define RFS_LAUNCH_COMBINE(SPH)                                                                                          \
        if (stage && lds_c > 64 * 1024) HIPCHK(c, hipFuncSetAttribute((const void*)k_swd_combine<SPH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c)); \
        hipLaunchKernelGGL(k_swd_combine<SPH>, dim3((nchain + 31) / 32), dim3(64, 8), lds_c,      /* 512-thread blocks = 16 layer slots per chain (round 5: 5.29 -> 5.24 ms per step against 256 threads; 1024 the same, 128 slower: 5.38); same sums in the same order */                             \
                           st, nchain, n, c->mode, nt, R, c->wt, c->mrf.as<double>(),                                    \
                           c->krn.as<double>(), c->croot.as<double>(), c->ugr.as<double>(), c->sflag.as<int>(),          \
                           P.nseq, c->d_dobs.as<double>(), misfit, grad, dsyn, flag,                                      \
                           track ? c->wvalid.as<int>() : (int*)nullptr, rowc, first, stage)
        if (c->sphere) { RFS_LAUNCH_COMBINE(true); } else { RFS_LAUNCH_COMBINE(false); }
#undef RFS_LAUNCH_COMBINE
        HIPCHK(c, hipGetLastError());
        return RFS_OK;
    };
    const bool early_combine = c->has_swd && c->has_rf && c->mode == 0 && !part && !tiled && !rf_time;
    if (c->has_swd && c->has_rf) {     // the latency-bound root search runs beside the RF kernels
        hipStream_t ss = part ? c->stream2m : ((warm && c->warm_serial) ? user : c->stream2);
        HIPCHK(c, hipEventRecord(c->ev_fork, user));
        HIPCHK(c, hipStreamWaitEvent(ss, c->ev_fork, 0));
        TRY(launch_swd(c, ss, nchain, n, P, !part, true, 0, 0, warm));
        // partitioned step with a Love block: the RF half carries the Love search ahead of its sweeps and ends last, so the
        // Rayleigh eigenfunction pass runs on the search half right behind the search instead of waiting for the join
        if (a_eigen) TRY(launch_swd(c, ss, nchain, n, P, true, false, 3));
        if (early_combine) { ENSURE(c, c->mrf, (size_t)nchain * sizeof(double)); TRY(launch_combine(ss, 1)); }
        HIPCHK(c, hipEventRecord(c->ev_join, ss));
    } else if (c->has_swd) {
        TRY(launch_swd(c, user, nchain, n, P, true, true, 0, 0, warm));
    }
    if (c->has_rf) {
        if (part) { HIPCHK(c, hipStreamWaitEvent(c->stream3, c->ev_fork, 0)); c->stream = c->stream3; }
        int rc = RFS_OK;
        if (rf_time) {
            { KTimer t(c, RFS_K_RF_PASS_A, c->stream); rc = launch_passA(c, nchain, n, c->f, true); }
            // model_rf.py:162-196 with method "time": rf and kernels from cal_rf_par_time_all; the gradient
            // K.r is accumulated spike by spike (rf_time_kernels.hpp), no kernel trace is materialised
            double* ds = dsyn;
            if (!ds) { rc = rc ? rc : ensure(c, c->dsyn, (size_t)nchain * c->ndata * sizeof(double)); ds = c->dsyn.as<double>(); }
            if (!rc) { KTimer t(c, RFS_K_RF_MID, c->stream);
                rc = rft_forward(c, nchain, c->f, ds, (size_t)c->ndata);
                if (!rc) rc = ensure(c, c->mrf, (size_t)nchain * sizeof(double));
                if (!rc) rc = ensure(c, c->Cres, (size_t)nchain * (c->f.nft / 2) * sizeof(double));
                if (!rc) {
                    size_t lds = ((size_t)c->f.nft + c->f.nt) * sizeof(double);
                    if (c->f.nft > 4096)
                        hipLaunchKernelGGL(k_rft_resid_cres_big, dim3(nchain), dim3(256), 0, c->stream, c->f, ds, c->ndata,
                                           c->d_dobs.as<double>(), c->pulse_ts.as<double>(), c->mrf.as<double>(),
                                           c->Cres.as<double>());
                    else
                    hipLaunchKernelGGL(k_rft_resid_cres, dim3(nchain), dim3(256), lds, c->stream, c->f, ds, c->ndata,
                                       c->d_dobs.as<double>(), c->pulse_ts.as<double>(), c->mrf.as<double>(),
                                       c->Cres.as<double>());
                } }
            if (!rc) { KTimer t(c, RFS_K_RF_PASS_B, c->stream);
                rc = ensure(c, c->PG, (size_t)nchain * 4 * n * sizeof(double));
                if (!rc) rc = rft_partials(c, nchain, n, c->f, c->Cres.as<double>(), c->PG.as<double>(), nullptr); }
        } else {
            // Frequency-domain pipeline, one chain tile at a time (a single tile unless the row scratch of the whole
            // batch would exceed rf_scratch_budget): pass A -> spectrum, IFFT, residual, FFT -> pass B -> RF part of
            // the gradient.  The reduction does not wait for the search, so it runs here, on the RF stream.
            for (int c0 = 0; c0 < nchain && !rc; c0 += rf_tile) {
                const int nc = std::min(rf_tile, nchain - c0);
                { KTimer t(c, RFS_K_RF_PASS_A, c->stream); rc = launch_passA(c, nc, n, c->f, true, (size_t)c0, rf_peel); }
                if (!rc) { KTimer t(c, RFS_K_RF_MID, c->stream);
                    rc = launch_mid(c, nc, n, c->f, c->d_dobs.as<double>(), c->ndata,
                                    dsyn ? dsyn + (size_t)c0 * c->ndata : nullptr, true, (size_t)c0, (size_t)nchain); }
                if (!rc) { KTimer t(c, RFS_K_RF_PASS_B, c->stream); rc = launch_passB(c, nc, n, c->f, (size_t)c0, rf_peel); }
                if (!rc) { {      // (outside the timed group: the join below is not pass B's time)
                        // (early_combine: one tile; the surface-wave part is in place once the join has passed)
                        if (early_combine) { if (hipStreamWaitEvent(c->stream, c->ev_join, 0) != hipSuccess) rc = RFS_ERR_HIP; }
                        const RfReduce rr{c->PG.as<double>(), c->mrf.as<double>() + c0, c->cr.as<double>() + (size_t)c0 * 2 * n,
                                          c->d_dobs.as<double>(), n, rf_nparts_b(c->f), (int)!c->has_swd, (int)early_combine, c->wt};
                        if (defer && !tiled && (early_combine || !c->has_swd)) *defer = rr;      // the caller's next kernel reduces (k_flow_post)
                        else hipLaunchKernelGGL(k_rf_reduce, dim3(nc), dim3(n <= 64 ? 64 : 128), 0, c->stream, nc, rr, misfit + c0,
                                                grad + (size_t)c0 * 2 * n, flag + c0,
                                                dsyn ? dsyn + (size_t)c0 * c->ndata : (double*)nullptr, c->ndata);
                    } }
            }
            rf_reduced = true;
            if (!rc && early_items > 0) rc = launch_swd(c, c->stream3, nchain, n, P, true, false, 1, early_items);
        }
        c->stream = user;
        if (rc) return rc;
        if (part) { HIPCHK(c, hipEventRecord(c->ev_join3, c->stream3)); HIPCHK(c, hipStreamWaitEvent(user, c->ev_join3, 0)); }
    }
    if (c->has_swd && c->has_rf && !early_combine) HIPCHK(c, hipStreamWaitEvent(user, c->ev_join, 0));      // (early_combine: joined before the reduction)
    if (part) TRY(launch_swd(c, user, nchain, n, P, true, false, a_eigen ? 4 : (early_items > 0 ? 2 : 0)));   // (rest of the) eigenfunction pass, whole chip
    if (c->has_rf && !rf_reduced) {
        KTimer t(c, RFS_K_COMBINE, c->stream);
        const RfReduce rr{c->PG.as<double>(), c->mrf.as<double>(), c->cr.as<double>(), nullptr, n, rf_time ? 1 : rf_nparts(c->f),
                          (int)!c->has_swd, 0, 1.0};
        hipLaunchKernelGGL(k_rf_reduce, dim3(nchain), dim3(n <= 64 ? 64 : 128), 0, c->stream, nchain, rr, misfit, grad, flag,
                           (double*)nullptr, 0);
        HIPCHK(c, hipGetLastError());
    }
    if (c->has_swd && !early_combine) TRY(launch_combine(c->stream, 0));
    if (timed) { HIPCHK(c, hipEventRecord(cal->ev[2 * cal->stage + 1], user)); cal->stage++; }
    if (track) { c->warm_primed = true; c->warm_nchain = nchain; c->warm_nitems = P.nitems; }
    return RFS_OK;
}

}  // namespace

extern "C" {

// The surface-wave stream carries the step's longest dependent chain (warm start -> branch test -> reference-root stage ->
// eigenfunctions -> combine); the receiver-function sweeps beside it have slack.  Where the device offers stream priorities
// the surface-wave stream gets the high one, so that its wavefronts are dispatched first when both streams have work
// (RFS_SWD_STREAM_PRIORITY=0 in the environment: plain stream, for A/B measurements).
static bool create_swd_stream(hipStream_t* s) {
    int lo = 0, hi = 0;
    const char* e = getenv("RFS_SWD_STREAM_PRIORITY");
    if ((!e || atoi(e) != 0) && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi < lo &&
        hipStreamCreateWithPriority(s, hipStreamNonBlocking, hi) == hipSuccess)
        return true;
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking) == hipSuccess;
}

int rfs_create(rfs_ctx** out, int device, int max_chains, int max_layers) {
    if (!out || max_chains < 1 || max_layers < 2) return RFS_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return RFS_ERR_HIP;
    rfs_ctx* c = new rfs_ctx();
    c->device = device; c->max_chains = max_chains; c->max_layers = max_layers;
    bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreate(&c->stream) == hipSuccess &&
              create_swd_stream(&c->stream2) &&
              hipStreamCreateWithFlags(&c->stream_l, hipStreamNonBlocking) == hipSuccess &&      // (plain priority; a high one for the background search: 4.6 -> 4.8 ms per step, round 6)

              hipEventCreateWithFlags(&c->ev_lf, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_lj, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_join3, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < RFS_BG_SLOTS; i++) ok = hipEventCreateWithFlags(&c->ev_bg[i], hipEventDisableTiming) == hipSuccess;
    // (RFS_WALK_STREAM=0 in the environment: the walk stays on the surface-wave stream -- A/B measurements.  Measured at the bench's
    // configuration with the default four hardware queues: 7.05 -> 6.47 ms per step; with GPU_MAX_HW_QUEUES = 6 / 8: 6.8)
    if (ok && !(getenv("RFS_WALK_STREAM") && atoi(getenv("RFS_WALK_STREAM")) == 0))
        ok = hipStreamCreateWithFlags(&c->stream_w, hipStreamNonBlocking) == hipSuccess &&      // (plain priority: with the surface-wave stream's high one the step is 5 % slower, round 6)
             hipEventCreateWithFlags(&c->ev_wk[0], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->ev_wk[1], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->ev_x1, hipEventDisableTiming) == hipSuccess;
    if (!ok) { delete c; return RFS_ERR_HIP; }
    c->own_stream = true;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete c; return RFS_ERR_HIP; }
    c->ncu = prop.multiProcessorCount;
    if (make_partition_streams(c) != RFS_OK) c->cu_split = 0;      // masks unsupported: fall back to shared CUs
    // (RFS_CTX_OPTS="name=value,name=value" in the environment: options every new context starts with -- for A/B runs of
    // scripts that build their contexts themselves; unknown names are reported on stderr and ignored)
    if (const char* e = getenv("RFS_CTX_OPTS")) {
        std::string all(e), applied;
        size_t p0 = 0;
        while (p0 < all.size()) {
            size_t p1 = all.find(',', p0);
            if (p1 == std::string::npos) p1 = all.size();
            const std::string kv = all.substr(p0, p1 - p0);
            const size_t eq = kv.find('=');
            char* endp = nullptr;
            const long v = eq != std::string::npos ? strtol(kv.c_str() + eq + 1, &endp, 10) : 0;
            if (kv.empty()) {}
            else if (eq == std::string::npos || eq == 0 || endp == kv.c_str() + eq + 1 || *endp != '\0')
                fprintf(stderr, "rfsurf: RFS_CTX_OPTS: malformed entry '%s' (want name=integer) -- ignored\n", kv.c_str());
            else if (rfs_set_option(c, kv.substr(0, eq).c_str(), (int)v) != RFS_OK)
                fprintf(stderr, "rfsurf: RFS_CTX_OPTS: %s not applied (%s)\n", kv.c_str(), c->err.c_str());
            else applied += (applied.empty() ? "" : ", ") + kv;
            p0 = p1 + 1;
        }
        // (these settings move parity-relevant behaviour: never silently)
        if (!applied.empty()) fprintf(stderr, "rfsurf: RFS_CTX_OPTS applied to a new context: %s\n", applied.c_str());
    }
    *out = c;
    return RFS_OK;
}

void rfs_destroy(rfs_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    Buf* bufs[] = {&c->d_tw[0], &c->d_tw[1], &c->d_tw[2], &c->d_tw[3], &c->mdlSR, &c->mdlL, &c->mdlcL, &c->sphR, &c->sphL, &c->d_dobs, &c->x, &c->misfit, &c->grad, &c->dsyn, &c->flag, &c->lc, &c->cr,
                   &c->d_minv, &c->spec3, &c->ts3, &c->S0f, &c->S0p, &c->pulse_spec, &c->pulse_ts, &c->Pbuf, &c->Cres,
                   &c->mdl, &c->RR, &c->Rs, &c->spec, &c->tser, &c->wres, &c->W, &c->wmax2, &c->PG, &c->mrf,
                   &c->croot, &c->sflag, &c->edone, &c->cds, &c->krn, &c->ugr, &c->b1a, &c->b1b, &c->b1c, &c->b1d, &c->b1e,
                   &c->b1f, &c->b1g, &c->specp, &c->tserp, &c->klbuf, &c->bt, &c->lx, &c->lp, &c->lU, &c->lgrad,
                   &c->ldsyn, &c->lflag, &c->mdlc, &c->xw, &c->dxT, &c->crT, &c->wvalid, &c->wneed, &c->wlist, &c->wforce,
                   &c->wstats, &c->wsgn, &c->crs, &c->craw, &c->wilist, &c->wlist2, &c->wlist3, &c->cwarm, &c->fpend, &c->twid, &c->gtab, &c->etab, &c->fstat, &c->RT, &c->wslope, &c->rstat, &c->wbetmx, &c->wsg1, &c->slist, &c->scount, &c->hi32, &c->stat32, &c->wferr, &c->wspA, &c->wspB, &c->wspc, &c->frec, &c->xsp, &c->xspc, &c->xredo, &c->Hs, &c->cold_roots, &c->cold_nroot, &c->cold_s0, &c->cold_ticket, &c->wcold};
    for (Buf* b : bufs) if (b->p) hipFree(b->p);
    if (c->h_wcount) hipHostFree(c->h_wcount);
    if (c->h_scount) hipHostFree(c->h_scount);
    for (auto e : c->ev_w) if (e) hipEventDestroy(e);
    for (auto e : c->ev_bg) if (e) hipEventDestroy(e);
    for (auto e : c->ev_wk) if (e) hipEventDestroy(e);
    if (c->ev_x1) hipEventDestroy(c->ev_x1);
    if (c->stream_w) hipStreamDestroy(c->stream_w);
    drop_plans(c);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    if (c->stream2) hipStreamDestroy(c->stream2);
    if (c->stream_l) hipStreamDestroy(c->stream_l);
    if (c->ev_lf) hipEventDestroy(c->ev_lf);
    if (c->ev_lj) hipEventDestroy(c->ev_lj);
    if (c->stream2m) hipStreamDestroy(c->stream2m);
    if (c->stream3) hipStreamDestroy(c->stream3);
    if (c->ev_join3) hipEventDestroy(c->ev_join3);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    for (auto& pool : c->tev) for (auto e : pool) hipEventDestroy(e);
    for (auto& kv : c->calib) for (auto e : kv.second.ev) if (e) hipEventDestroy(e);
    delete c;
}

const char* rfs_last_error(const rfs_ctx* c) { return c ? c->err.c_str() : "null context"; }

int rfs_set_stream(rfs_ctx* c, void* s) {
    if (!c) return RFS_ERR_ARG;
    if (c->own_stream && c->stream) { hipStreamSynchronize(c->stream); hipStreamDestroy(c->stream); }
    c->stream = (hipStream_t)s; c->own_stream = false;
    return RFS_OK;
}

int rfs_synchronize(rfs_ctx* c) {
    if (!c) return RFS_ERR_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream2));
    if (c->stream_l) HIPCHK(c, hipStreamSynchronize(c->stream_l));
    if (c->stream_w) HIPCHK(c, hipStreamSynchronize(c->stream_w));
    if (c->stream2m) HIPCHK(c, hipStreamSynchronize(c->stream2m));
    if (c->stream3) HIPCHK(c, hipStreamSynchronize(c->stream3));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RFS_OK;
}

int rfs_enable_timing(rfs_ctx* c, int on) {
    if (!c) return RFS_ERR_ARG;
    c->timing = on != 0;
    c->timing_mask = (on == 0 || on == 1) ? ~0u : ((unsigned)on >> 1);      // on = 1: all groups; on = 2 * mask: those groups only
    for (auto& u : c->tused) u = 0;
    return RFS_OK;
}

int rfs_kernel_ms_sum(rfs_ctx* c, double* ms, int32_t* count) {
    if (!c || !ms || !count) return RFS_ERR_ARG;
    TRY(rfs_synchronize(c));
    for (int i = 0; i < RFS_K_COUNT; i++) {
        double tot = 0.0;
        for (size_t k = 0; k + 1 < c->tused[i]; k += 2) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, c->tev[i][k], c->tev[i][k + 1]) == hipSuccess) tot += t;
        }
        ms[i] = tot; count[i] = (int32_t)(c->tused[i] / 2);
        c->tused[i] = 0;
    }
    return RFS_OK;
}

int rfs_kernel_timeline(rfs_ctx* c, double* start_ms, double* end_ms, int32_t* count) {
    if (!c || !start_ms || !end_ms || !count) return RFS_ERR_ARG;
    TRY(rfs_synchronize(c));
    for (int i = 0; i < RFS_K_COUNT; i++) {
        double a = 0.0, b = 0.0; int nn = 0;
        for (size_t k = 0; k + 1 < c->tused[i]; k += 2) {
            hipEvent_t ref = k / 2 < c->tref[i].size() ? c->tref[i][k / 2] : nullptr;
            float t0 = 0.f, t1 = 0.f;
            if (!ref) continue;
            if (hipEventElapsedTime(&t0, ref, c->tev[i][k]) != hipSuccess || hipEventElapsedTime(&t1, ref, c->tev[i][k + 1]) != hipSuccess) continue;
            a += t0; b += t1; nn++;
        }
        start_ms[i] = a; end_ms[i] = b; count[i] = nn;
    }
    return RFS_OK;
}

int rfs_ndata(const rfs_ctx* c) { return c ? c->ndata : 0; }



int rfs_set_option(rfs_ctx* c, const char* name, int value) {
    if (!c || !name) return RFS_ERR_ARG;
    if (!strcmp(name, "share_rc_rg")) {      // takes effect at the next rfs_joint_setup
        c->share_rc_rg = value != 0; if (!c->share_rc_rg) c->rg_alias = c->lg_alias = false; return RFS_OK;
    }
    if (!strcmp(name, "swd_speculate")) {
        if (value != -1 && value != 1 && value != 2 && value != 4) return fail(c, RFS_ERR_ARG, "swd_speculate must be -1, 1, 2 or 4");
        c->swd_speculate = value; return RFS_OK;
    }
    if (!strcmp(name, "swd_segments")) {
        if (value != -1 && value != 1 && value != 2 && value != 4) return fail(c, RFS_ERR_ARG, "swd_segments must be -1, 1, 2 or 4");
        c->swd_segments = value; return RFS_OK;
    }
    if (!strcmp(name, "swd_lanes_per_chain")) {
        if (value != 0 && (value < 1 || value > 64 || (value & (value - 1)))) return fail(c, RFS_ERR_ARG, "swd_lanes_per_chain must be 0 or a power of two <= 64");
        c->swd_lanes = value; return RFS_OK;
    }
    if (!strcmp(name, "rf_scratch_budget_mb")) {
        if (value < 1) return fail(c, RFS_ERR_ARG, "rf_scratch_budget_mb must be positive");
        c->rf_scratch_budget = (size_t)value << 20; return RFS_OK;
    }
    if (!strcmp(name, "rf_band_limit_digits")) {
        if (value < 0 || value > 300) return fail(c, RFS_ERR_ARG, "rf_band_limit_digits must be within [0, 300]");
        c->rf_band_digits = value;
        if (c->configured && c->has_rf) { HIPCHK(c, hipSetDevice(c->device)); TRY(rfs_synchronize(c)); set_band_limit(c->f, value, c->rf_band_floor); }
        for (auto& kv : c->calib) kv.second.stage = -1;
        return RFS_OK;
    }
    if (!strcmp(name, "swd_walk_dense")) {
        if (value < 0 || value > 1) return fail(c, RFS_ERR_ARG, "swd_walk_dense must be 0 or 1");
        c->walk_dense = value; return RFS_OK;
    }
    if (!strcmp(name, "rf_mid_fused")) {
        if (value < 0 || value > 1) return fail(c, RFS_ERR_ARG, "rf_mid_fused must be 0 or 1");
        c->rf_mid_fused = value; return RFS_OK;
    }
    if (!strcmp(name, "rf_f32_beyond_band")) {
        if (value < 0 || value > 1) return fail(c, RFS_ERR_ARG, "rf_f32_beyond_band must be 0 or 1");
        c->rf_f32 = value; return RFS_OK;
    }
    if (!strcmp(name, "rf_peel_check")) {
        if (value < 0 || value > 1) return fail(c, RFS_ERR_ARG, "rf_peel_check must be 0 or 1");
        c->rf_peel_check = value;
        if (c->rstat.p) { HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, hipMemsetAsync(c->rstat.p, 0, sizeof(unsigned), c->stream)); }
        return RFS_OK;
    }
    if (!strcmp(name, "rf_row_peeling")) {
        if (value < -1 || value > 2) return fail(c, RFS_ERR_ARG, "rf_row_peeling must be -1 / 1 (per chain, where safe), 0 (never) or 2 (always)");
        c->rf_peel = value;
        for (auto& kv : c->calib) kv.second.stage = -1;
        return RFS_OK;
    }
    if (!strcmp(name, "rf_band_floor_digits")) {
        if (value < 0 || value > 300) return fail(c, RFS_ERR_ARG, "rf_band_floor_digits must be within [0, 300]");
        c->rf_band_floor = value;
        if (c->configured && c->has_rf) { HIPCHK(c, hipSetDevice(c->device)); TRY(rfs_synchronize(c)); set_band_limit(c->f, c->rf_band_digits, value); }
        for (auto& kv : c->calib) kv.second.stage = -1;
        return RFS_OK;
    }
    if (!strcmp(name, "swd_warm_start")) {
        if (value < 0 || value > 2) return fail(c, RFS_ERR_ARG, "swd_warm_start must be 0, 1 or 2");
        c->warm_opt = value; c->warm_primed = false; return RFS_OK;
    }
    if (!strcmp(name, "swd_warm_exact")) {
        if (value < 0 || value > 1) return fail(c, RFS_ERR_ARG, "swd_warm_exact must be 0 or 1");
        c->warm_exact = value; c->warm_primed = false; return RFS_OK;
    }
    if (!strcmp(name, "swd_exact_group_small")) {
        if (value < 1 || value > 4096) return fail(c, RFS_ERR_ARG, "swd_exact_group_small must be within [1, 4096]");
        c->exact_group_small = value; return RFS_OK;
    }
    if (!strcmp(name, "swd_exact_group")) {
        if (value < 1 || value > 4096) return fail(c, RFS_ERR_ARG, "swd_exact_group must be within [1, 4096]");
        c->exact_group = value; return RFS_OK;      // (1: the 16-lane form only; a lane per group takes >= 2)
    }
    if (!strcmp(name, "swd_walk_window")) {
        if (value < -1 || value > 4096) return fail(c, RFS_ERR_ARG, "swd_walk_window must be -1 or within [0, 4096]");
        c->walk_window = value; return RFS_OK;
    }
    if (!strcmp(name, "rf_store_hyp")) { c->rf_store_hyp = value != 0; return RFS_OK; }
    if (!strcmp(name, "swd_exact_redo_runup")) {
        if (value < -1 || value > 64) return fail(c, RFS_ERR_ARG, "swd_exact_redo_runup must be within [-1, 64]");
        c->exact_redo_runup = value; return RFS_OK;
    }
    if (!strcmp(name, "swd_exact_overlap")) { c->exact_overlap = value != 0; return RFS_OK; }
    if (!strcmp(name, "swd_exact_budget")) {
        if (value < 0 || value > 100000) return fail(c, RFS_ERR_ARG, "swd_exact_budget must be within [0, 100000]");
        c->exact_budget = value; return RFS_OK;
    }
    if (!strcmp(name, "swd_cold_again")) { c->cold_again = value != 0; return RFS_OK; }
    if (!strcmp(name, "swd_cold_first")) {
        if (value < 0 || value > COLD_MAX_CHAINS) return fail(c, RFS_ERR_ARG, "swd_cold_first must be within [0, 512]");
        c->cold_first = value; return RFS_OK;
    }
    if (!strcmp(name, "swd_cold_scan")) {
        if (value < -1 || value > 1) return fail(c, RFS_ERR_ARG, "swd_cold_scan must be -1, 0 or 1");
        c->cold_scan = value; return RFS_OK;
    }
    if (!strcmp(name, "swd_exact_coop")) {
        if (value < 0 || value > 2) return fail(c, RFS_ERR_ARG, "swd_exact_coop must be 0, 1 or 2");
        c->exact_coop = value; return RFS_OK;
    }
    if (!strcmp(name, "swd_exact_origin_tol_e9")) {
        if (value < 0 || value > 2000) return fail(c, RFS_ERR_ARG, "swd_exact_origin_tol_e9 must be within [0, 2000] (units of 1e-9 c)");
        c->exact_origin_tol = (float)(value * 1.0e-9); return RFS_OK;
    }
    if (!strcmp(name, "swd_exact_runup")) {
        if (value < 0 || value > 64) return fail(c, RFS_ERR_ARG, "swd_exact_runup must be within [0, 64]");
        c->exact_runup = value; return RFS_OK;
    }
    if (!strcmp(name, "flow_async_handback")) {
        if (value < 0 || value > 1) return fail(c, RFS_ERR_ARG, "flow_async_handback must be 0 or 1");
        HIPCHK(c, hipSetDevice(c->device));
        TRY(rfs_synchronize(c));
        c->flow_async = value; for (auto& b : c->bg_busy) b = false;
        c->flow_x = nullptr; c->warm_primed = false;
        if (c->fpend.p) HIPCHK(c, hipMemset(c->fpend.p, 0, c->fpend.cap));
        return RFS_OK;
    }
    if (!strcmp(name, "swd_warm_serial")) { c->warm_serial = value != 0; return RFS_OK; }
    if (!strcmp(name, "swd_warm_widen")) { c->warm_widen = value != 0; return RFS_OK; }
    if (!strcmp(name, "swd_warm_feedback")) { c->warm_feedback = value != 0; return RFS_OK; }
    if (!strcmp(name, "swd_warm_last_round_coop")) { c->warm_coop = value != 0; return RFS_OK; }
    if (!strcmp(name, "swd_warm_round_budgets")) {
        if (value < 0 || value > 999999) return fail(c, RFS_ERR_ARG, "swd_warm_round_budgets must be b1 + 100 b2 + 10000 b3 with 0 <= b < 100");
        // a later round without the one before it never runs (the launches chain b1 -> b2 -> b3): say so instead of dropping it
        const int b1 = value % 100, b2 = (value / 100) % 100, b3 = (value / 10000) % 100;
        if ((b1 == 0 && (b2 > 0 || b3 > 0)) || (b2 == 0 && b3 > 0))
            return fail(c, RFS_ERR_ARG, "swd_warm_round_budgets: a round needs the one before it (b1 = 0 with b2 > 0, or b2 = 0 with b3 > 0)");
        c->warm_budgets = (int)value; return RFS_OK;
    }
    if (!strcmp(name, "flow_skip_idle")) { c->flow_skip_idle = value != 0; return RFS_OK; }
    if (!strcmp(name, "swd_warm_reset")) { c->warm_primed = false; return RFS_OK; }      // next evaluation: full search
    if (!strcmp(name, "swd_exact_final")) {
        c->exact_final = value != 0;
        if (c->wforce.p) { HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, hipMemsetAsync(c->wforce.p, 0, c->wforce.cap, c->stream)); }
        return RFS_OK;
    }
    if (!strcmp(name, "recalibrate")) { for (auto& kv : c->calib) kv.second.stage = -1; return RFS_OK; }
    if (!strcmp(name, "cu_split")) {
        if (value < 0 || value > 2) return fail(c, RFS_ERR_ARG, "cu_split must be 0, 1 or 2");
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipDeviceSynchronize());
        c->cu_split = value;
        for (auto& kv : c->calib) kv.second.stage = -1;
        return make_partition_streams(c);
    }
    if (!strcmp(name, "early_eigen_periods")) {
        if (value < -1) return fail(c, RFS_ERR_ARG, "early_eigen_periods must be -1 (automatic), 0 (off) or a period count");
        c->early_eigen = value; return RFS_OK;
    }
    return fail(c, RFS_ERR_ARG, std::string("unknown option ") + name);
}

int rfs_get_stat(rfs_ctx* c, const char* name, int64_t* value) {
    if (!c || !name || !value) return RFS_ERR_ARG;
    *value = 0;
    int idx = -1;
    if (!strcmp(name, "rf_peel_residual")) {          // largest closure residual of the row peeling since rf_peel_check was set, in units of 1e-18
        if (!c->rstat.p) return RFS_OK;
        HIPCHK(c, hipSetDevice(c->device));
        TRY(rfs_synchronize(c));
        unsigned u = 0; float q;
        HIPCHK(c, hipMemcpy(&u, c->rstat.p, sizeof(u), hipMemcpyDeviceToHost));
        std::memcpy(&q, &u, sizeof(q));
        *value = (q > 9.0e0f) ? INT64_MAX : (int64_t)((double)q * 1.0e18);
        return RFS_OK;
    }
    if (!strcmp(name, "rf_band_bins") || !strcmp(name, "rf_bins")) {       // frequencies inside the gradient's band / all of them (n2)
        *value = c->has_rf ? (!strcmp(name, "rf_bins") ? c->f.n2 : std::min(c->f.nk, c->f.n2)) : 0;
        return RFS_OK;
    }
    if (!strcmp(name, "rf_f32_chains") || !strcmp(name, "rf_f32_resweeps")) {    // (chain, evaluation) pairs swept in float32 beyond the band / swept again in f64
        if (!c->stat32.p) return RFS_OK;
        HIPCHK(c, hipSetDevice(c->device));
        TRY(rfs_synchronize(c));
        unsigned long long v[66];
        HIPCHK(c, hipMemcpy(v, c->stat32.p, sizeof(v), hipMemcpyDeviceToHost));
        if (!strcmp(name, "rf_f32_resweeps")) *value = (int64_t)v[0];
        else for (int i = 2; i < 66; i++) *value += (int64_t)v[i];
        return RFS_OK;
    }
    if (!strcmp(name, "flow_chain_steps")) {          // (chain, step) pairs the flow entries advanced a trajectory by
        if (!c->fstat.p) return RFS_OK;
        HIPCHK(c, hipSetDevice(c->device));
        TRY(rfs_synchronize(c));
        unsigned long long v[64];
        HIPCHK(c, hipMemcpy(v, c->fstat.p, sizeof(v), hipMemcpyDeviceToHost));
        for (int i = 0; i < 64; i++) *value += (int64_t)v[i];
        return RFS_OK;
    }
    if (!strncmp(name, "wstat_", 6)) { idx = atoi(name + 6); if (idx < 0 || idx > 39) idx = -1; }      // (raw slot of the warm start's counters)
    else if (!strcmp(name, "swd_warm_declined_chains")) idx = 0;
    else if (!strcmp(name, "swd_warm_secular_evals")) idx = 1;
    else if (!strcmp(name, "swd_warm_items")) idx = 2;
    else if (!strcmp(name, "swd_warm_walked_chains")) idx = 12;
    else if (!strcmp(name, "swd_warm_wide_chains")) idx = 13;
    else if (!strcmp(name, "swd_exact_declined_chains")) idx = 14;
    else if (!strcmp(name, "swd_exact_secular_evals")) idx = 15;
    else if (!strcmp(name, "swd_exact_evals_slowest_lane")) idx = 3;
    else if (!strcmp(name, "swd_exact_wavefronts")) idx = 16;
    else if (!strncmp(name, "swd_warm_cause_", 15)) { idx = atoi(name + 15); if (idx < 4 || idx > 11) idx = -1; }
    else if (!strcmp(name, "swd_warm_fail_no_change")) idx = 24;
    else if (!strcmp(name, "swd_warm_fail_other")) idx = 25;
    else if (!strcmp(name, "swd_warm_search_evals")) idx = 26;
    else if (!strcmp(name, "swd_warm_search_evals_slowest_lane")) idx = 27;
    else if (!strcmp(name, "swd_warm_search_lanes")) idx = 28;
    else if (!strcmp(name, "swd_cold_chains")) idx = 32;
    else if (!strcmp(name, "swd_cold_secular_evals")) idx = 33;
    else if (!strncmp(name, "swd_cold_fail_", 14)) { idx = atoi(name + 14); idx = (idx < 34 || idx > 39) ? -1 : idx; }
    else if (!strncmp(name, "swd_warm_passed_on_", 19)) { idx = atoi(name + 19); idx = (idx < 1 || idx > 3) ? -1 : 28 + idx; }
    else if (!strncmp(name, "swd_exact_cause_", 16)) { idx = atoi(name + 16); idx = (idx < 1 || idx > 7) ? -1 : 16 + idx; }
    if (idx < 0) return fail(c, RFS_ERR_ARG, std::string("unknown statistic ") + name);
    if (!c->wstats.p) return RFS_OK;
    HIPCHK(c, hipSetDevice(c->device));
    TRY(rfs_synchronize(c));
    unsigned long long v[40];
    HIPCHK(c, hipMemcpy(v, c->wstats.p, sizeof(v), hipMemcpyDeviceToHost));
    *value = (int64_t)v[idx];
    return RFS_OK;
}

// The roots (phase velocities, float32 values as the reference stores them) of the LAST evaluation, as they lie in the
// persistent root buffer: diagnostics for parity tests that want to know which chains of a flow step hold a root that is
// not the sequential search's.  c: HOST [nchain][nitems]; returns the number of items through *nitems.
int rfs_last_roots(rfs_ctx* c, int nchain, int32_t* nitems, double* croots) {
    if (!c || !nitems) return RFS_ERR_ARG;
    *nitems = c->warm_nitems;
    if (!croots) return RFS_OK;
    if (!c->configured || !c->croot.p || c->warm_nitems <= 0 || nchain != c->warm_nchain ||
        c->croot.cap < (size_t)c->warm_nitems * nchain * sizeof(double))
        return fail(c, RFS_ERR_STATE, "rfs_last_roots: no evaluation of that many chains yet");
    HIPCHK(c, hipSetDevice(c->device));
    TRY(rfs_synchronize(c));
    std::vector<double> tmp((size_t)c->warm_nitems * nchain);
    HIPCHK(c, hipMemcpy(tmp.data(), c->croot.p, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int e = 0; e < c->warm_nitems; e++)
        for (int k = 0; k < nchain; k++) croots[(size_t)k * c->warm_nitems + e] = tmp[(size_t)e * nchain + k];
    return RFS_OK;
}

// ---------------------------------------------------------------- B1 / libsurf
static int swd_b1(rfs_ctx* c, int nchain, int nlayer, const double* thk, const double* vp, const double* vs,
                  const double* rho, int nper, const double* period, int wavetype, int mode, int sphere,
                  bool kernels, double* cout, double* dcda, double* dcdb, double* dcdr, double* dcdh, int32_t* flag) {
    TRY(check_batch(c, nchain, nlayer));
    if (wavetype < RFS_WAVE_RC || wavetype > RFS_WAVE_LG) return fail(c, RFS_ERR_ARG, "wavetype should be one of [Rc,Rg,Lc,Lg]");
    if (mode < 0 || mode > 64) return fail(c, RFS_ERR_ARG, "mode must be within [0, 64]");
    sphere = sphere ? 1 : 0;
    if (nper < 1 || !period || !thk || !vp || !vs || !rho || !cout || !flag) return fail(c, RFS_ERR_ARG, "null/empty argument");
    HIPCHK(c, hipSetDevice(c->device));
    const int n = nlayer;
    size_t mb = (size_t)nchain * n * sizeof(double);
    TRY(upload(c, c->b1a, thk, mb)); TRY(upload(c, c->b1b, vp, mb));
    TRY(upload(c, c->b1c, vs, mb)); TRY(upload(c, c->b1d, rho, mb));
    TRY(upload(c, c->bt, period, (size_t)nper * sizeof(double)));
    ENSURE(c, c->mdl, (size_t)4 * n * nchain * sizeof(float));
    ENSURE(c, c->mdlc, (size_t)6 * n * nchain * sizeof(double));
    int nth = nchain * n;
    hipLaunchKernelGGL(k_prep_swd_b1, dim3((nth + 255) / 256), dim3(256), 0, c->stream, nchain, n, c->b1a.as<double>(),
                       c->b1b.as<double>(), c->b1c.as<double>(), c->b1d.as<double>(), c->mdl.as<float>(),
                       c->mdlc.as<double>());
    const bool rg = (wavetype & 1) != 0;       // group velocity: _RayleighGroup / _LoveGroup (surfdisp.cpp:119-173)
    int ntw[4] = {0, 0, 0, 0};
    const double* tw[4] = {nullptr, nullptr, nullptr, nullptr};
    ntw[wavetype] = nper; tw[wavetype] = c->bt.as<double>();
    size_t nn = (size_t)n * nchain;
    if (sphere) ENSURE(c, (wavetype < 2) ? c->sphR : c->sphL, 7 * nn * sizeof(double));
    // forward: roots at T only (+ U from sregn96 / slegn96 for the group types); kernel: three passes for groups
    SwdPlan P = make_plan(ntw, tw, kernels, sphere, kernels ? 0 : 1, !kernels, c->sphR.as<double>(), c->sphL.as<double>());
    TRY(launch_family_prep(c, c->stream, nchain, n, P, sphere));
    c->warm_primed = false;
    c->swd_mode_cur = mode;
    c->swd_water_cur = false;
    for (int ch = 0; ch < nchain; ch++) c->swd_water_cur = c->swd_water_cur || (float)vs[(size_t)ch * n] <= 0.0f;
    const int rc_swd = launch_swd(c, c->stream, nchain, n, P, kernels || rg);
    c->swd_mode_cur = 0; c->swd_water_cur = false;
    TRY(rc_swd);
    size_t cb = (size_t)nchain * nper * sizeof(double), kb = cb * n;
    ENSURE(c, c->b1e, cb);
    double* dk[4] = {nullptr, nullptr, nullptr, nullptr};
    if (kernels) {
        ENSURE(c, c->klbuf, 4 * kb);
        for (int i = 0; i < 4; i++) dk[i] = c->klbuf.as<double>() + (size_t)i * nchain * nper * n;
    }
    int ng = nchain * nper;
    ENSURE(c, c->krn, 8); ENSURE(c, c->ugr, 8);
    hipLaunchKernelGGL(sphere ? k_swd_export<true> : k_swd_export<false>, dim3((ng + 127) / 128), dim3(128), 0, c->stream, nchain, n, P.R,
                       c->krn.as<double>(), c->croot.as<double>(), c->ugr.as<double>(), c->b1e.as<double>(),
                       dk[0], dk[1], dk[2], dk[3]);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<int> sf((size_t)P.nseq * nchain);
    HIPCHK(c, hipMemcpy(sf.data(), c->sflag.p, sf.size() * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(cout, c->b1e.p, cb, hipMemcpyDeviceToHost));
    if (kernels) {
        HIPCHK(c, hipMemcpy(dcda, dk[0], kb, hipMemcpyDeviceToHost)); HIPCHK(c, hipMemcpy(dcdb, dk[1], kb, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(dcdr, dk[2], kb, hipMemcpyDeviceToHost)); HIPCHK(c, hipMemcpy(dcdh, dk[3], kb, hipMemcpyDeviceToHost));
    }
    for (int ch = 0; ch < nchain; ch++) {
        int ok = 1;
        for (int s = 0; s < P.nseq; s++) ok = ok && sf[(size_t)s * nchain + ch];
        flag[ch] = ok;
        if (!ok && rg) for (int k = 0; k < nper; k++) cout[(size_t)ch * nper + k] = 0.0;
    }
    return RFS_OK;
}

int rfs_swd_forward(rfs_ctx* c, int nchain, int nlayer, const double* thk, const double* vp, const double* vs,
                    const double* rho, int nper, const double* period, int wavetype, int mode, int sphere,
                    double* cout, int32_t* flag) {
    return swd_b1(c, nchain, nlayer, thk, vp, vs, rho, nper, period, wavetype, mode, sphere, false, cout,
                  nullptr, nullptr, nullptr, nullptr, flag);
}

int rfs_swd_kernel(rfs_ctx* c, int nchain, int nlayer, const double* thk, const double* vp, const double* vs,
                   const double* rho, int nper, const double* period, int wavetype, int mode, int sphere,
                   double* cout, double* dcda, double* dcdb, double* dcdr, double* dcdh, int32_t* flag) {
    if (!dcda || !dcdb || !dcdr || !dcdh) return fail(c, RFS_ERR_ARG, "null kernel output");
    return swd_b1(c, nchain, nlayer, thk, vp, vs, rho, nper, period, wavetype, mode, sphere, true, cout,
                  dcda, dcdb, dcdr, dcdh, flag);
}

// ---------------------------------------------------------------- B1 / librf
static int rf_b1_tile(rfs_ctx* c, int nchain, int nlayer, const double* thk, const double* rho, const double* vp,
                      const double* vs, const double* qa, const double* qb, const rfs_rf_params* par, double* rf, double* kl);

// Host-pointer entry: any number of chains, in tiles (one grid row per chain in the RF sweeps: <= 32768 per launch; with
// kernels the 4 n partial spectra + traces of a tile stay within ~6 GB) -- every buffer is tile-local, the results of a
// chain do not depend on the tiling.
static int rf_b1(rfs_ctx* c, int nchain, int nlayer, const double* thk, const double* rho, const double* vp,
                 const double* vs, const double* qa, const double* qb, const rfs_rf_params* par, double* rf, double* kl) {
    TRY(check_batch(c, nchain, nlayer));
    TRY(check_rf(c, par));
    if (!thk || !rho || !vp || !vs || !qa || !qb || !rf) return fail(c, RFS_ERR_ARG, "null argument");
    const size_t n = (size_t)nlayer, nft = (size_t)rf_nextpow2(par->nt);
    size_t tile = RF_MAX_CHAINS_PER_LAUNCH;
    if (kl) {
        const size_t per_chain = 4 * n * ((nft / 2 + 1) * sizeof(cplx) + nft * sizeof(double) + (size_t)par->nt * sizeof(double));
        tile = std::min(tile, std::max<size_t>(64, (size_t)6e9 / per_chain / 64 * 64));
    }
    for (size_t c0 = 0; c0 < (size_t)nchain; c0 += tile) {
        const int nc = (int)std::min(tile, (size_t)nchain - c0);
        const size_t o = c0 * n;
        TRY(rf_b1_tile(c, nc, nlayer, thk + o, rho + o, vp + o, vs + o, qa + o, qb + o, par, rf + c0 * (size_t)par->nt,
                       kl ? kl + c0 * 4 * n * (size_t)par->nt : nullptr));
    }
    return RFS_OK;
}

static int rf_b1_tile(rfs_ctx* c, int nchain, int nlayer, const double* thk, const double* rho, const double* vp,
                      const double* vs, const double* qa, const double* qb, const rfs_rf_params* par, double* rf, double* kl) {
    HIPCHK(c, hipSetDevice(c->device));
    const int n = nlayer;
    RfFreq f = make_freq(*par, kl ? 0 : 1);
    size_t mb = (size_t)nchain * n * sizeof(double);
    TRY(upload(c, c->b1a, thk, mb)); TRY(upload(c, c->b1b, rho, mb)); TRY(upload(c, c->b1c, vp, mb));
    TRY(upload(c, c->b1d, vs, mb)); TRY(upload(c, c->b1e, qa, mb)); TRY(upload(c, c->b1f, qb, mb));
    ENSURE(c, c->lc, (size_t)nchain * n * sizeof(RfLayer));
    int nth = nchain * n;
    hipLaunchKernelGGL(k_prep_rf_b1, dim3((nth + 255) / 256), dim3(256), 0, c->stream, nchain, n, c->b1a.as<double>(),
                       c->b1b.as<double>(), c->b1c.as<double>(), c->b1d.as<double>(), c->b1e.as<double>(),
                       c->b1f.as<double>(), f.p, c->lc.as<RfLayer>());
    TRY(launch_passA(c, nchain, n, f, kl != nullptr));
    ENSURE(c, c->b1g, (size_t)nchain * f.nt * sizeof(double));
    if (f.method != RFS_RF_FREQ) {       // time domain: iterative deconvolution
        TRY(rft_forward(c, nchain, f, c->b1g.as<double>(), (size_t)f.nt));
        if (kl) {
            ENSURE(c, c->klbuf, (size_t)nchain * 4 * n * f.nt * sizeof(double));
            TRY(rft_partials(c, nchain, n, f, nullptr, nullptr, c->klbuf.as<double>()));
        }
    } else {
    TRY(launch_mid(c, nchain, n, f, nullptr, f.nt, c->b1g.as<double>(), false));
    if (kl) {
        size_t ntr = (size_t)nchain * 4 * n;
        ENSURE(c, c->specp, ntr * f.n2 * sizeof(cplx));
        ENSURE(c, c->tserp, ntr * f.nft * sizeof(double));
        ENSURE(c, c->klbuf, ntr * f.nt * sizeof(double));
        dim3 grid(rf_chunks(f), nchain);
        hipLaunchKernelGGL(k_rf_partial_spectra<false>, grid, dim3(rf_block(f)), 0, c->stream, nchain, n, f,
                           c->lc.as<RfLayer>(), c->RR.as<double>(), c->Rs.as<double>(), c->wmax2.as<double>(),
                           c->specp.as<cplx>());
        hipLaunchKernelGGL(k_rf_partial_spectra<true>, dim3((nchain + 63) / 64), dim3(64), 0, c->stream, nchain, n, f,
                           c->lc.as<RfLayer>(), c->RR.as<double>(), c->Rs.as<double>(), c->wmax2.as<double>(),
                           c->specp.as<cplx>());
        HIPCHK(c, hipGetLastError());
        TRY(run_fft(c, f.nft, ntr, 1, c->specp.p, c->tserp.p));
        size_t tot = ntr * f.nt;
        hipLaunchKernelGGL(k_rf_scale_kl, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, ntr, f,
                           c->tserp.as<double>(), c->klbuf.as<double>());
        HIPCHK(c, hipGetLastError());
    }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(rf, c->b1g.p, (size_t)nchain * f.nt * sizeof(double), hipMemcpyDeviceToHost));
    if (kl) HIPCHK(c, hipMemcpy(kl, c->klbuf.p, (size_t)nchain * 4 * n * f.nt * sizeof(double), hipMemcpyDeviceToHost));
    return RFS_OK;
}

int rfs_rf_forward(rfs_ctx* c, int nchain, int nlayer, const double* thk, const double* rho, const double* vp,
                   const double* vs, const double* qa, const double* qb, const rfs_rf_params* par, double* rf) {
    return rf_b1(c, nchain, nlayer, thk, rho, vp, vs, qa, qb, par, rf, nullptr);
}

int rfs_rf_kernel_all(rfs_ctx* c, int nchain, int nlayer, const double* thk, const double* rho, const double* vp,
                      const double* vs, const double* qa, const double* qb, const rfs_rf_params* par, double* rf,
                      double* kl) {
    if (!kl) return fail(c, RFS_ERR_ARG, "null kernel output");
    return rf_b1(c, nchain, nlayer, thk, rho, vp, vs, qa, qb, par, rf, kl);
}

// ---------------------------------------------------------------- B2 / plugins
int rfs_joint_setup2(rfs_ctx* c, int nlayer, const rfs_rf_params* rf, const rfs_swd_params* swd, double sigma1,
                     double sigma2, const double* dobs) {
    if (!c) return RFS_ERR_ARG;
    c->configured = false;
    c->warm_primed = false; c->flow_x = nullptr;
    if (nlayer < 2 || nlayer > c->max_layers || nlayer > MAXL) return fail(c, RFS_ERR_ARG, "nlayer outside [2, min(max_layers,128)]");
    int ntw[4] = {0, 0, 0, 0};
    const double* tw[4] = {nullptr, nullptr, nullptr, nullptr};
    int sphere = 0;
    if (swd) {
        ntw[0] = swd->ntRc; ntw[1] = swd->ntRg; ntw[2] = swd->ntLc; ntw[3] = swd->ntLg;
        tw[0] = swd->tRc; tw[1] = swd->tRg; tw[2] = swd->tLc; tw[3] = swd->tLg;
        if (swd->mode < 0 || swd->mode > 64) return fail(c, RFS_ERR_ARG, "mode must be within [0, 64]");
        sphere = swd->sphere ? 1 : 0;
    }
    int nswd = 0;
    for (int i = 0; i < 4; i++) {
        if (ntw[i] < 0 || (ntw[i] > 0 && !tw[i])) return fail(c, RFS_ERR_ARG, "bad period lists");
        nswd += ntw[i];
    }
    if (!rf && nswd == 0) return fail(c, RFS_ERR_ARG, "neither RF nor SWD data configured");
    if (rf) TRY(check_rf(c, rf));
    HIPCHK(c, hipSetDevice(c->device));
    c->n = nlayer; c->has_rf = rf != nullptr; c->has_swd = nswd > 0;
    c->swd_mode = swd ? swd->mode : 0;
    c->mode = (c->has_rf && c->has_swd) ? 0 : (c->has_rf ? 1 : 2);
    c->sphere = sphere;
    c->has_minv = false;
    int nt = 0;
    if (rf) { c->f = make_freq(*rf, 0); set_band_limit(c->f, c->rf_band_digits, c->rf_band_floor); nt = rf->nt; }
    c->ndata = nt + nswd;
    // wt = (sigma1/sigma2)^2 n1/n2, model_rf_swd_vs_thk.py:79
    c->wt = c->has_swd && c->has_rf ? (sigma1 / sigma2) * (sigma1 / sigma2) * nt / (double)nswd : 1.0;
    for (int i = 0; i < 4; i++) {
        c->ntw[i] = ntw[i];
        if (ntw[i]) TRY(upload(c, c->d_tw[i], tw[i], (size_t)ntw[i] * sizeof(double)));
    }
    c->rg_alias = c->share_rc_rg && ntw[0] > 0 && ntw[0] == ntw[1] && memcmp(tw[0], tw[1], (size_t)ntw[0] * sizeof(double)) == 0;
    c->lg_alias = c->share_rc_rg && ntw[2] > 0 && ntw[2] == ntw[3] && memcmp(tw[2], tw[3], (size_t)ntw[2] * sizeof(double)) == 0;
    ENSURE(c, c->d_dobs, (size_t)c->ndata * sizeof(double));
    if (dobs) HIPCHK(c, hipMemcpyAsync(c->d_dobs.p, dobs, (size_t)c->ndata * sizeof(double), hipMemcpyHostToDevice, c->stream));
    else HIPCHK(c, hipMemsetAsync(c->d_dobs.p, 0, (size_t)c->ndata * sizeof(double), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // dummies so that unused pointers are never null-dereferenced in combine
    ENSURE(c, c->PG, 8); ENSURE(c, c->mrf, 8); ENSURE(c, c->krn, 8); ENSURE(c, c->croot, 8); ENSURE(c, c->ugr, 8);
    ENSURE(c, c->sflag, 16);
    c->configured = true;
    return RFS_OK;
}

int rfs_joint_setup(rfs_ctx* c, int nlayer, const rfs_rf_params* rf, int ntRc, const double* tRc, int ntRg,
                    const double* tRg, double sigma1, double sigma2, const double* dobs) {
    rfs_swd_params swd{};
    swd.ntRc = ntRc; swd.tRc = tRc; swd.ntRg = ntRg; swd.tRg = tRg;
    return rfs_joint_setup2(c, nlayer, rf, (ntRc > 0 || ntRg > 0) ? &swd : nullptr, sigma1, sigma2, dobs);
}

int rfs_joint_misfit_grad_dev(rfs_ctx* c, int nchain, const double* x, double* misfit, double* grad, double* dsyn,
                              int32_t* flag) {
    if (!c) return RFS_ERR_ARG;
    if (!c->configured) return fail(c, RFS_ERR_STATE, "rfs_joint_setup has not been called");
    TRY(check_batch(c, nchain, c->n));
    if (!x || !misfit || !grad || !flag) return fail(c, RFS_ERR_ARG, "null argument");
    return joint_eval(c, nchain, x, misfit, grad, dsyn, flag);
}

int rfs_joint_misfit_grad(rfs_ctx* c, int nchain, const double* x, double* misfit, double* grad, double* dsyn,
                          int32_t* flag) {
    if (!c) return RFS_ERR_ARG;
    if (!c->configured) return fail(c, RFS_ERR_STATE, "rfs_joint_setup has not been called");
    TRY(check_batch(c, nchain, c->n));
    if (!x || !misfit || !grad || !flag) return fail(c, RFS_ERR_ARG, "null argument");
    const int n = c->n;
    TRY(upload(c, c->x, x, (size_t)nchain * 2 * n * sizeof(double)));
    ENSURE(c, c->misfit, (size_t)nchain * sizeof(double)); ENSURE(c, c->grad, (size_t)nchain * 2 * n * sizeof(double));
    ENSURE(c, c->dsyn, (size_t)nchain * c->ndata * sizeof(double)); ENSURE(c, c->flag, (size_t)nchain * sizeof(int));
    TRY(joint_eval(c, nchain, c->x.as<double>(), c->misfit.as<double>(), c->grad.as<double>(), c->dsyn.as<double>(),
                   c->flag.as<int>()));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(misfit, c->misfit.p, (size_t)nchain * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(grad, c->grad.p, (size_t)nchain * 2 * n * sizeof(double), hipMemcpyDeviceToHost));
    if (dsyn) HIPCHK(c, hipMemcpy(dsyn, c->dsyn.p, (size_t)nchain * c->ndata * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(flag, c->flag.p, (size_t)nchain * sizeof(int), hipMemcpyDeviceToHost));
    return RFS_OK;
}

int rfs_joint_forward(rfs_ctx* c, int nchain, const double* x, int quirk, double* dsyn, int32_t* flag) {
    if (!c) return RFS_ERR_ARG;
    if (!c->configured) return fail(c, RFS_ERR_STATE, "rfs_joint_setup has not been called");
    TRY(check_batch(c, nchain, c->n));
    if (!x || !dsyn || !flag) return fail(c, RFS_ERR_ARG, "null argument");
    HIPCHK(c, hipSetDevice(c->device));
    const int n = c->n;
    TRY(upload(c, c->x, x, (size_t)nchain * 2 * n * sizeof(double)));
    ENSURE(c, c->dsyn, (size_t)nchain * c->ndata * sizeof(double)); ENSURE(c, c->flag, (size_t)nchain * sizeof(int));
    ENSURE(c, c->cr, (size_t)nchain * 2 * n * sizeof(double));
    if (c->has_rf) ENSURE(c, c->lc, (size_t)nchain * n * sizeof(RfLayer));
    if (c->has_swd) { ENSURE(c, c->mdl, (size_t)4 * n * nchain * sizeof(float)); ENSURE(c, c->mdlc, (size_t)6 * n * nchain * sizeof(double)); }
    int nth = nchain * n;
    hipLaunchKernelGGL(k_prep_joint, dim3((nth + 255) / 256), dim3(256), 0, c->stream, nchain, n, c->x.as<double>(),
                       (int)c->has_rf, c->f.p, c->lc.as<RfLayer>(), c->cr.as<double>(), (int)c->has_swd, c->mdl.as<float>(),
                       c->mdlc.as<double>(), (double*)nullptr, (size_t)0, (int*)nullptr, (size_t)0,
                       (double*)nullptr, (double*)nullptr, (double*)nullptr, FlowPre{});
    c->warm_primed = false;                    // croot / krn are about to be overwritten by an unrelated evaluation
    int nt = c->has_rf ? c->f.nt : 0;
    if (c->has_rf) {
        RfFreq f = c->f; f.fwd_order = 1; f.pi64 = 0;      // cal_rf_freq / cal_rf_time frequency axis
        TRY(launch_passA(c, nchain, n, f, false));
        if (f.method != RFS_RF_FREQ) TRY(rft_forward(c, nchain, f, c->dsyn.as<double>(), (size_t)c->ndata));
        else TRY(launch_mid(c, nchain, n, f, nullptr, c->ndata, c->dsyn.as<double>(), false));
    }
    if (c->has_swd) {
        // model_surf.py:104-131 computes every block at tRc; quirk != 0 keeps that (the reference then needs
        // every block to have len(tRc) rows, otherwise its slice assignment raises)
        const double* tw[4]; int ntw[4];
        for (int i = 0; i < 4; i++) {
            ntw[i] = c->ntw[i];
            tw[i] = c->d_tw[i].as<double>();
            if (quirk && c->ntw[0] > 0 && ntw[i] > 0) {
                if (ntw[i] != c->ntw[0])
                    return fail(c, RFS_ERR_UNSUPPORTED, "forward quirk needs every block as long as tRc, as the reference does");
                tw[i] = c->d_tw[0].as<double>();
            }
        }
        size_t nn = (size_t)n * nchain;
        if (c->sphere && ntw[0] + ntw[1] > 0) ENSURE(c, c->sphR, 7 * nn * sizeof(double));
        if (c->sphere && ntw[2] + ntw[3] > 0) ENSURE(c, c->sphL, 7 * nn * sizeof(double));
        SwdPlan P = make_plan(ntw, tw, false, c->sphere, 1, true, c->sphR.as<double>(), c->sphL.as<double>(),
                              c->share_rc_rg && (c->rg_alias || (quirk && c->ntw[0] > 0)));
        TRY(launch_family_prep(c, c->stream, nchain, n, P, c->sphere));
        c->swd_mode_cur = c->swd_mode;
        const int rc_swd = launch_swd(c, c->stream, nchain, n, P, ntw[1] + ntw[3] > 0);
        c->swd_mode_cur = 0;
        TRY(rc_swd);
        ENSURE(c, c->ugr, 8);
        hipLaunchKernelGGL(k_swd_forward_out, dim3((nchain + 63) / 64), dim3(64), 0, c->stream, nchain, nt, P.R,
                           c->croot.as<double>(), c->ugr.as<double>(), c->sflag.as<int>(), P.nseq,
                           c->dsyn.as<double>(), c->flag.as<int>());
    } else {
        HIPCHK(c, hipMemsetAsync(c->flag.p, 0, (size_t)nchain * sizeof(int), c->stream));
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(dsyn, c->dsyn.p, (size_t)nchain * c->ndata * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(flag, c->flag.p, (size_t)nchain * sizeof(int), hipMemcpyDeviceToHost));
    if (!c->has_swd) for (int i = 0; i < nchain; i++) flag[i] = 1;
    return RFS_OK;
}

// ---------------------------------------------------------------- leapfrog
int rfs_leapfrog_dev2(rfs_ctx* c, int nchain, const double* x0, const double* p0, const double* dt, const int32_t* L,
                      int32_t Lmax, const int32_t* nactive, const double* bounds, double* xnew, double* Ucur,
                      double* Unew, double* Hcur, double* Hnew, double* dsyn_cur, double* dsyn_new, int32_t* ok) {
    if (!c) return RFS_ERR_ARG;
    if (!c->configured) return fail(c, RFS_ERR_STATE, "rfs_joint_setup has not been called");
    TRY(check_batch(c, nchain, c->n));
    if (!x0 || !p0 || !dt || !L || !bounds || !xnew || !Ucur || !Unew || !Hcur || !Hnew || !dsyn_cur || !dsyn_new || !ok || Lmax < 1)
        return fail(c, RFS_ERR_ARG, "null argument");
    if (nactive)
        for (int s = 0; s < Lmax; s++)
            if (nactive[s] < 1 || nactive[s] > nchain || (s > 0 && nactive[s] > nactive[s - 1]))
                return fail(c, RFS_ERR_ARG, "nactive must be non-increasing within [1, nchain]");
    const int n = c->n, nx = 2 * n, nd = c->ndata;
    ENSURE(c, c->lx, (size_t)nchain * nx * sizeof(double)); ENSURE(c, c->lp, (size_t)nchain * nx * sizeof(double));
    ENSURE(c, c->lU, (size_t)nchain * sizeof(double)); ENSURE(c, c->lgrad, (size_t)nchain * nx * sizeof(double));
    ENSURE(c, c->ldsyn, (size_t)nchain * nd * sizeof(double)); ENSURE(c, c->lflag, (size_t)nchain * sizeof(int));
    double *x = c->lx.as<double>(), *p = c->lp.as<double>(), *U = c->lU.as<double>(), *g = c->lgrad.as<double>(),
           *d = c->ldsyn.as<double>();
    int* fl = c->lflag.as<int>();
    const double* minv = c->has_minv ? c->d_minv.as<double>() : nullptr;
    // The start models go through the reference-semantics search whatever the options say: a batch of trajectories is
    // then a function of its arguments alone (a resumed run repeats an uninterrupted one bit for bit), and Ucur / Hcur
    // are the reference's numbers.  The steps continue from there.  (nactive: a shrinking chain count changes the layout
    // of the kept roots / kernels -- no warm start.)
    const int traj = nactive == nullptr ? 1 : 0;
    int* wforce = nullptr;
    if (traj && c->exact_final && c->has_swd && c->warm_opt) {
        ENSURE(c, c->wforce, (size_t)nchain * sizeof(int));
        wforce = c->wforce.as<int>();
    }
    TRY(joint_eval(c, nchain, x0, U, g, d, fl, traj ? 2 : 0));
    hipLaunchKernelGGL(k_leap_begin, dim3(nchain), dim3(64), 0, c->stream, nchain, nx, nd, minv, x0, p0, dt, L, Lmax, U, g, d, fl,
                       x, p, Ucur, Hcur, Unew, Hnew, dsyn_cur, dsyn_new, ok);
    // failed chains (ok = 0: failed evaluation, or L outside [1, Lmax]) keep xnew = x0, Hnew = +inf written by
    // k_leap_begin (reference returns (xcur, inf, dobs, False))
    HIPCHK(c, hipMemcpyAsync(xnew, x0, (size_t)nchain * nx * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    for (int step = 0; step < Lmax; step++) {
        // chains sorted by decreasing L: only the first nactive[step] are still inside their trajectory
        const int na = nactive ? nactive[step] : nchain;
        const int nth = na * nx;
        hipLaunchKernelGGL(k_leap_drift, dim3((nth + 255) / 256), dim3(256), 0, c->stream, na, nx, step, minv, dt, L, bounds, x, p, ok,
                           wforce);
        TRY(joint_eval(c, na, x, U, g, d, fl, traj));
        hipLaunchKernelGGL(k_leap_kick, dim3(na), dim3(64), 0, c->stream, na, nx, nd, step, minv, dt, L, x, U, g, d, fl, p,
                           Unew, Hnew, dsyn_new, xnew, ok);
    }
    HIPCHK(c, hipGetLastError());
    return RFS_OK;
}

int rfs_set_inverse_mass(rfs_ctx* c, const double* minv) {
    if (!c) return RFS_ERR_ARG;
    if (!c->configured) return fail(c, RFS_ERR_STATE, "rfs_joint_setup has not been called");
    if (!minv) { c->has_minv = false; return RFS_OK; }
    for (int i = 0; i < 2 * c->n; i++)
        if (!(minv[i] > 0.0) || minv[i] > 1.0e300) return fail(c, RFS_ERR_ARG, "inverse masses must be positive and finite");
    HIPCHK(c, hipSetDevice(c->device));
    TRY(upload(c, c->d_minv, minv, (size_t)2 * c->n * sizeof(double)));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->has_minv = true;
    return RFS_OK;
}

int rfs_flow_step2(rfs_ctx* c, int nchain, double* x, double* p, const double* dt, int32_t* rem, int32_t* fresh,
                   const double* bounds, double* Ucur, double* Hcur, double* Unew, double* Hnew, double* dsyn_cur,
                   double* dsyn_new, int32_t* ok, int32_t* done, const rfs_flow_next* next) {
    return rfs_flow_step3(c, nchain, x, p, dt, rem, fresh, bounds, Ucur, Hcur, Unew, Hnew, dsyn_cur, dsyn_new, ok, done, next, nullptr);
}

int rfs_flow_step3(rfs_ctx* c, int nchain, double* x, double* p, const double* dt, int32_t* rem, int32_t* fresh,
                   const double* bounds, double* Ucur, double* Hcur, double* Unew, double* Hnew, double* dsyn_cur,
                   double* dsyn_new, int32_t* ok, int32_t* done, const rfs_flow_next* next, const rfs_flow_records* records) {
    if (!c) return RFS_ERR_ARG;
    if (!c->configured) return fail(c, RFS_ERR_STATE, "rfs_joint_setup has not been called");
    TRY(check_batch(c, nchain, c->n));
    if (!x || !p || !dt || !rem || !fresh || !bounds || !Ucur || !Hcur || !Unew || !Hnew || !dsyn_cur || !dsyn_new || !ok || !done)
        return fail(c, RFS_ERR_ARG, "null argument");
    FlowNext fn{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    if (next) {
        if (!next->have || !next->u || !next->p || !next->xstart || !next->res_x || !next->res_val)
            return fail(c, RFS_ERR_ARG, "rfs_flow_next: only rem (with gsave, kick given), res_dsyn, gsave and kick may be null");
        if (!next->rem && (!next->gsave || !next->kick))
            return fail(c, RFS_ERR_ARG, "rfs_flow_next: rem == NULL (length and step size follow later) needs gsave and kick");
        if ((next->gsave == nullptr) != (next->kick == nullptr))
            return fail(c, RFS_ERR_ARG, "rfs_flow_next: gsave and kick go together");
        fn = FlowNext{next->have, next->u, next->p, next->rem, next->xstart, next->res_x, next->res_val, next->res_dsyn,
                      next->gsave, next->kick, nullptr, nullptr, nullptr, 0};
    }
    const int n = c->n, nx = 2 * n, nd = c->ndata;
    ENSURE(c, c->lU, (size_t)nchain * sizeof(double)); ENSURE(c, c->lgrad, (size_t)nchain * nx * sizeof(double));
    ENSURE(c, c->ldsyn, (size_t)nchain * nd * sizeof(double)); ENSURE(c, c->lflag, (size_t)nchain * sizeof(int));
    double *U = c->lU.as<double>(), *g = c->lgrad.as<double>(), *d = c->ldsyn.as<double>();
    int* fl = c->lflag.as<int>();
    const int nth = nchain * nx;
    const double* minv = c->has_minv ? c->d_minv.as<double>() : nullptr;
    int* wforce = nullptr;
    if (c->exact_final && c->has_swd && c->warm_opt) { ENSURE(c, c->wforce, (size_t)nchain * sizeof(int)); wforce = c->wforce.as<int>(); }
    // The previous evaluation (roots, kernels, model) and the pending flags describe ONE set of chains: the state whose x
    // array the last flow step advanced.  Another state on the same context starts from the full search -- and with chains
    // sitting steps out there is no way to serve two states in turn.
    if (c->flow_x != x) {
        if (c->flow_async && c->flow_x) {
            bool busy = false;
            for (bool b : c->bg_busy) busy = busy || b;
            if (busy) return fail(c, RFS_ERR_STATE, "flow_async_handback: a context advances one flow state at a time (another x array while searches of the "
                                                    "previous one are outstanding); use one context per state or switch the option off");
        }
        c->warm_primed = false;
        if (c->fpend.p) HIPCHK(c, hipMemsetAsync(c->fpend.p, 0, c->fpend.cap, c->stream));
        c->flow_x = x;
    }
    // chains a step hands back to the full search may sit that step out ("flow_async_handback"): who did is kept here
    const size_t pend_before = c->fpend.cap;
    ENSURE(c, c->fpend, (size_t)nchain * sizeof(int));
    if (c->fpend.cap != pend_before || c->fpend_nchain != nchain) {
        HIPCHK(c, hipMemsetAsync(c->fpend.p, 0, c->fpend.cap, c->stream));
        c->fpend_nchain = nchain;
    }
    const FlowPre fpre{minv, dt, rem, fresh, ok, bounds, x, p, fn.gsave, fn.kick, wforce, c->fpend.as<int>()};
    (void)nth;
    RfReduce rr{};
    KTimer tstep(c, RFS_K_FLOW_STEP, c->stream);      // the whole step on the caller's stream: first launch .. behind k_flow_post
    struct StepEvGuard { rfs_ctx* c; ~StepEvGuard() { c->cur_step_ev = nullptr; } } step_ev_guard{c};
    c->flow_cur = true; c->f_rem = rem; c->f_fresh = fresh; c->f_ok = ok;
    const int rc_eval = joint_eval(c, nchain, x, U, g, d, fl, 1, &fpre, &rr);      // (drift with mirror reflection inside k_prep_joint; RF reduction left to k_flow_post)
    c->flow_cur = false; c->f_rem = c->f_fresh = c->f_ok = nullptr;
    TRY(rc_eval);
    const int* need_cur = c->last_async ? c->wneed.as<int>() + (size_t)c->wpar * (3 * (size_t)nchain + 4) : (const int*)nullptr;
    if (next && c->has_swd && c->warm_opt && c->warm_primed && c->warm_nchain == nchain && c->xw.p) {
        // the start roots of every running trajectory, restored with the start model when it is rejected (k_flow_post)
        const int nitems = (int)(c->croot.cap / sizeof(double) / (size_t)nchain);
        const int nit = c->ntw[0] + c->ntw[1] + c->ntw[2] + c->ntw[3] > 0 ? c->warm_nitems : 0;
        if (nit > 0 && nit <= nitems) {
            ENSURE(c, c->crs, (size_t)nit * nchain * sizeof(double));
            fn.croot = c->croot.as<double>(); fn.crs = c->crs.as<double>(); fn.xw = c->xw.as<double>(); fn.nitems = nit;
        }
    }
    if (!c->fstat.p) {
        ENSURE(c, c->fstat, 64 * sizeof(unsigned long long));
        HIPCHK(c, hipMemsetAsync(c->fstat.p, 0, 64 * sizeof(unsigned long long), c->stream));
    }
    FlowRec frec{nullptr, nullptr, 0, 0, 0, 0.0};
    if (records && records->buf) {
        const int stride = 8 + nx + (records->want_dsyn ? nd : 0);
        if (records->cap < 2 * nchain || records->bytes < (uint64_t)records->cap * stride * sizeof(double))
            return fail(c, RFS_ERR_ARG, "rfs_flow_records: a ring of cap >= 2 nchain records of 8 + 2 nlayer [+ ndata] doubles");
        if (!(records->stamp != 0.0)) return fail(c, RFS_ERR_ARG, "rfs_flow_records: stamp must not be 0 (what an empty slot holds)");
        if (!c->frec.p || records->reset) {
            ENSURE(c, c->frec, sizeof(unsigned long long));
            HIPCHK(c, hipMemsetAsync(c->frec.p, 0, sizeof(unsigned long long), c->stream));
        }
        frec = FlowRec{(double*)records->buf, c->frec.as<unsigned long long>(), records->cap, stride, records->want_dsyn ? 1 : 0, records->stamp};
    }
    hipLaunchKernelGGL(k_flow_post, dim3(nchain), dim3(64), 0, c->stream, nchain, nx, nd, minv, dt, x, U, g, d, fl, p, rem, fresh,
                       Ucur, Hcur, Unew, Hnew, dsyn_cur, dsyn_new, ok, done, fn, c->fstat.as<unsigned long long>(), rr,
                       need_cur, c->fpend.as<int>(), (int)c->wpar + 1, c->bg_ready, frec);
    HIPCHK(c, hipGetLastError());
    return RFS_OK;
}

int rfs_flow_restart(rfs_ctx* c, int nchain, int n1, const int32_t* idx1, const double* xkeep, int n2, const int32_t* idx2,
                     const double* pnew, const int32_t* remnew, const double* dtnew, int n3, const int32_t* idx3,
                     double* x, double* p, int32_t* rem, double* dt, int32_t* fresh, int32_t* ok, int32_t* nxt_have) {
    if (!c) return RFS_ERR_ARG;
    if (!c->configured) return fail(c, RFS_ERR_STATE, "rfs_joint_setup has not been called");
    TRY(check_batch(c, nchain, c->n));
    if (n1 < 0 || n2 < 0 || n3 < 0 || n1 > nchain || n2 > nchain || n3 > nchain) return fail(c, RFS_ERR_ARG, "list lengths must be within [0, nchain]");
    if ((n1 && (!idx1 || !xkeep || !x)) || (n2 && (!idx2 || !remnew || !rem)) || (n2 && pnew && (!p || !fresh || !ok)) ||
        (n2 && dtnew && !dt) || (n3 && (!idx3 || !nxt_have)))
        return fail(c, RFS_ERR_ARG, "null argument");
    if (n1 + n2 + n3 == 0) return RFS_OK;
    const int nx = 2 * c->n;
    hipLaunchKernelGGL(k_flow_restart, dim3(n1 + n2 + n3), dim3(64), 0, c->stream, nx, n1, n2, n3, idx1, xkeep, idx2, pnew, remnew,
                       dtnew, idx3, x, p, rem, dt, fresh, ok, nxt_have);
    HIPCHK(c, hipGetLastError());
    return RFS_OK;
}

int rfs_flow_deposit(rfs_ctx* c, void* hip_stream, int nchain, int n, const int32_t* idx, const double* u, const double* pnew,
                     const int32_t* remnew, const rfs_flow_next* next) {
    if (!c) return RFS_ERR_ARG;
    if (!c->configured) return fail(c, RFS_ERR_STATE, "rfs_joint_setup has not been called");
    TRY(check_batch(c, nchain, c->n));
    if (n < 0 || n > nchain) return fail(c, RFS_ERR_ARG, "list length must be within [0, nchain]");
    if (n == 0) return RFS_OK;
    if (!idx || !u || !pnew || !next || !next->have || !next->u || !next->p || (remnew && !next->rem))
        return fail(c, RFS_ERR_ARG, "null argument");
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    hipLaunchKernelGGL(k_flow_deposit, dim3(n), dim3(64), 0, st, 2 * c->n, n, idx, u, pnew, remnew, next->have,
                       const_cast<double*>(next->u), const_cast<double*>(next->p), const_cast<int*>(next->rem));
    HIPCHK(c, hipGetLastError());
    return RFS_OK;
}

int rfs_flow_step(rfs_ctx* c, int nchain, double* x, double* p, const double* dt, int32_t* rem, int32_t* fresh,
                  const double* bounds, double* Ucur, double* Hcur, double* Unew, double* Hnew, double* dsyn_cur,
                  double* dsyn_new, int32_t* ok, int32_t* done) {
    return rfs_flow_step2(c, nchain, x, p, dt, rem, fresh, bounds, Ucur, Hcur, Unew, Hnew, dsyn_cur, dsyn_new, ok, done, nullptr);
}

int rfs_leapfrog_dev(rfs_ctx* c, int nchain, const double* x0, const double* p0, const double* dt, const int32_t* L,
                     int32_t Lmax, const double* bounds, double* xnew, double* Ucur, double* Unew, double* Hcur,
                     double* Hnew, double* dsyn_cur, double* dsyn_new, int32_t* ok) {
    return rfs_leapfrog_dev2(c, nchain, x0, p0, dt, L, Lmax, nullptr, bounds, xnew, Ucur, Unew, Hcur, Hnew, dsyn_cur,
                             dsyn_new, ok);
}

}  // extern "C"
