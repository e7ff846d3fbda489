/*
 * rfsurf.h -- C ABI of librfsurf_hip.so: the MI355X (gfx950) implementation of the
 * misfit + gradient hot path of nqdu/RfSurfHmc.
 *
 * Every entry point names the reference interface it replaces (file:line under the
 * reference tree).  Conventions:
 *   - plain C, no exceptions; every call returns 0 on success or a negative rfs_status;
 *     rfs_last_error() gives the message.  Nothing ever calls exit() (the reference's
 *     bindings do: src/SWD/main.cpp:24, src/RF/main.cpp:40,162).
 *   - re-entrant: all state lives in the rfs_ctx (the reference Fortran keeps module
 *     globals and SAVEd variables, sregn96.f90:61-95, surfdisp96.f:423).
 *   - batched: one call evaluates `nchain` independent models ("chains").
 *   - arrays are contiguous, chain-major, float64 unless stated; "host" entry points take
 *     host pointers and copy; "_dev" entry points take DEVICE pointers and enqueue on the
 *     context's stream without synchronising.
 *   - there is NO CPU fallback: if no gfx950 device is usable rfs_create fails.
 */
#ifndef RFSURF_H
#define RFSURF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rfs_ctx rfs_ctx;

typedef enum {
    RFS_OK = 0,
    RFS_ERR_ARG = -1,         /* bad argument (also: unsupported enum value) */
    RFS_ERR_HIP = -2,         /* HIP runtime / rocFFT failure */
    RFS_ERR_STATE = -3,       /* call out of order (e.g. joint eval before joint setup) */
    RFS_ERR_UNSUPPORTED = -4  /* a request this library declines (see rfs_last_error; e.g. the forward quirk with unequal period blocks) */
} rfs_status;

/* wavetype codes of libsurf (src/SWD/main.cpp:17-24): strings "Rc","Rg","Lc","Lg" */
enum { RFS_WAVE_RC = 0, RFS_WAVE_RG = 1, RFS_WAVE_LC = 2, RFS_WAVE_LG = 3 };
/* rf_type codes (src/RF/main.cpp:28-41): "P"/"p" -> 1, "S"/"s" -> 2 */
enum { RFS_RF_P = 1, RFS_RF_S = 2 };
/* method codes (src/RF/main.cpp:44,52): "time" -> 0, anything else -> 1 (frequency domain) */
/* method: "time" = iterative time-domain deconvolution (deconit.f90), "freq" = water-level spectral division.
 * RFS_RF_TIME_PAR is "time" on the frequency axis of the single-parameter entry librf.kernel (cal_rf_par_time,
 * RFModule.f90:27,47: float32 pi) -- kernel_all (cal_rf_par_time_all :96,112) uses the f64 pi. */
enum { RFS_RF_TIME = 0, RFS_RF_FREQ = 1, RFS_RF_TIME_PAR = 2 };

/* Receiver-function scalars: the arguments of librf.forward / kernel_all
 * (src/RF/main.cpp:17-22, 140-146) and the keys of param.yaml's `rf:` block. */
typedef struct {
    double ray_p;       /* s/km */
    int32_t nt;         /* samples returned; FFT length = next power of two (deconit.f90:1-13) */
    double dt;          /* s */
    double gauss;       /* Gaussian f0 */
    double time_shift;  /* s; negated internally for rf_type S (main.cpp:35) */
    double water;       /* water level */
    int32_t rf_type;    /* RFS_RF_P / RFS_RF_S */
    int32_t method;     /* RFS_RF_TIME / RFS_RF_FREQ (/ RFS_RF_TIME_PAR); water is unused by the time method */
} rfs_rf_params;

/* -------- lifetime ------------------------------------------------------------------ */
/* device: HIP device ordinal (>= 0).  max_chains / max_layers size the workspaces. */
int rfs_create(rfs_ctx** ctx, int device, int max_chains, int max_layers);
void rfs_destroy(rfs_ctx* ctx);
const char* rfs_last_error(const rfs_ctx* ctx);
/* Use an existing HIP stream (e.g. torch.cuda.current_stream().cuda_stream) for all work. */
int rfs_set_stream(rfs_ctx* ctx, void* hip_stream);
/* Block until everything enqueued by this context has finished. */
int rfs_synchronize(rfs_ctx* ctx);

/* -------- B1: what libsurf exports (src/SWD/main.cpp:86-93) --------------------------- */
/* libsurf.forward(thk,vp,vs,rho,period,wavetype,mode,sphere) -> (c[nper], bool)
 * (src/SWD/main.cpp:14-59; _surfdisp surfdisp.cpp:62-109; _LoveGroup :119-141; _RayleighGroup :151-173).
 * All four wavetypes; sphere != 0 applies the earth-flattening transformation (surfdisp96.f:495-564 for the
 * root search, bldsph / sprayl / splove for group velocities) and returns spherical velocities
 * (_flat2sphere surfdisp.cpp:16-49).  mode: 0 fundamental, k > 0 the k-th higher mode -- the reference's mode loop
 * (surfdisp96.f:227-316: mode k searches above mode k-1 period by period; a mode that does not exist from some period on
 * gives c = 0 there WITHOUT clearing the flag, :337-362; only a failing fundamental does) runs inside the search's state
 * machine, bit for bit.
 * Model arrays [nchain][nlayer] are rounded to float32 first, as the binding does (main.cpp:9).
 * c: [nchain][nper]; flag[chain] = 1 ok, 0 root search failed (ierr == 1); the c values of a failed chain are
 * unspecified (the reference returns whatever roots it found before giving up, surfdisp.cpp:93-100).
 * "Lg" searches with vp = 1.732 vs as _LoveGroup does (it ignores the caller's vp).
 * Water layer: vs[0] <= 0 marks the top layer as a fluid (surfdisp96.f:138-139); the search then takes the reference's
 * water branch (:201-206 start value, :870-886 the fluid layer; Love: the layers below it, :750).  A fluid layer anywhere
 * else is as undefined here as in the reference (division by vs). */
int rfs_swd_forward(rfs_ctx* ctx, int nchain, int nlayer, const double* thk, const double* vp,
                    const double* vs, const double* rho, int nper, const double* period,
                    int wavetype, int mode, int sphere, double* c, int32_t* flag);
/* libsurf.adjoint_kernel(...) -> (c, dcda, dcdb, dcdr, dcdh, bool)
 * (src/SWD/main.cpp:61-82; _SurfKernel surfdisp.cpp:190-297; sregn96 / sregnpu
 * sregn96.f90:1637-1888; slegn96 / slegnpu slegn96.f90:672-919).  Kernel arrays: [nchain][nper][nlayer].
 * Love types: dcda is written as zeros (the reference leaves that array uninitialised, surfdisp.cpp:258-296).
 * Water layer on top (vs[0] <= 0): Rayleigh kernels from the fluid branches of sregn96 (sregn96.f90:555-575, 778-811,
 * 858-877, 931-945, 1122-1140, 1245-1262, 1456-1509); dcdb of the water layer is 0 (never assigned there); Love
 * kernels and Love group velocities are NaN, as the reference's are (slegn96 reads elements it never set). */
int rfs_swd_kernel(rfs_ctx* ctx, int nchain, int nlayer, const double* thk, const double* vp,
                   const double* vs, const double* rho, int nper, const double* period,
                   int wavetype, int mode, int sphere, double* c, double* dcda, double* dcdb,
                   double* dcdr, double* dcdh, int32_t* flag);

/* -------- B1: what librf exports (src/RF/main.cpp:193-212) ---------------------------- */
/* librf.forward(thk,rho,vp,vs,qa,qb,ray_p,nt,dt,gauss,time_shift,method,water,rf_type)
 * -> rf[nt]   (src/RF/main.cpp:17-62 -> cal_rf_freq RFModule.f90:193-255).  rf: [nchain][nt] */
int rfs_rf_forward(rfs_ctx* ctx, int nchain, int nlayer, const double* thk, const double* rho,
                   const double* vp, const double* vs, const double* qa, const double* qb,
                   const rfs_rf_params* par, double* rf);
/* librf.kernel_all(...) -> (rf[nt], k[4][nlayer][nt]), parameter axis = [rho, vp, vs, thk]
 * (src/RF/main.cpp:140-189 -> cal_rf_par_freq_all RFModule.f90:343-430).
 * kl: [nchain][4][nlayer][nt] */
int rfs_rf_kernel_all(rfs_ctx* ctx, int nchain, int nlayer, const double* thk, const double* rho,
                      const double* vp, const double* vs, const double* qa, const double* qb,
                      const rfs_rf_params* par, double* rf, double* kl);

/* -------- B2: the model-plugin hot path ----------------------------------------------- */
/* Configure what Joint_RF_SWD / ReceiverFunc / SurfWD hold (model/model_rf_swd_vs_thk.py:6-25,
 * model/model_rf.py:5-30, model/model_surf.py:5-45): RF scalars (rf == NULL: SWD-only plugin),
 * Rayleigh phase / group period lists (both 0: RF-only plugin), sigma1, sigma2, and the observed
 * data dobs[nt + ntRc + ntRg] = [rf, Rc, Rg] (host pointer, copied). */
int rfs_joint_setup(rfs_ctx* ctx, int nlayer, const rfs_rf_params* rf, int ntRc, const double* tRc,
                    int ntRg, const double* tRg, double sigma1, double sigma2, const double* dobs);
/* Same with the full SurfWD state (model/model_surf.py:5-29): the four period blocks Rc, Rg, Lc, Lg (data order
 * of dobs / dsyn), sphere, mode.  t* are the periods each block is EVALUATED at in misfit_and_grad: the
 * reference evaluates its Lc and Lg blocks at tRc (model_surf.py:199-216) -- a caller that wants that passes
 * tRc there.  swd == NULL: RF-only plugin. */
typedef struct {
    int32_t ntRc, ntRg, ntLc, ntLg;
    const double *tRc, *tRg, *tLc, *tLg;
    int32_t sphere;     /* 0 flat earth, != 0 earth flattening */
    int32_t mode;       /* 0 = fundamental, k > 0 = k-th higher mode (SurfWD(mode = ...), model_surf.py:5-7) */
} rfs_swd_params;
int rfs_joint_setup2(rfs_ctx* ctx, int nlayer, const rfs_rf_params* rf, const rfs_swd_params* swd,
                     double sigma1, double sigma2, const double* dobs);
/* misfit_and_grad(x) for nchain models at once (model_rf_swd_vs_thk.py:66-86, model_rf.py:137-198,
 * model_surf.py:155-228).  x: [nchain][2*nlayer] = [vs(0..n-1), thk(0..n-1)].
 * Outputs: misfit[nchain], grad[nchain][2*nlayer], dsyn[nchain][ndata], flag[nchain]
 * (flag 0 reproduces the plugins' failure returns: misfit 0, grad 0, dsyn = dobs for the joint
 * plugin / zeros for the SWD-only plugin).  DEVICE pointers; asynchronous on the ctx stream. */
int rfs_joint_misfit_grad_dev(rfs_ctx* ctx, int nchain, const double* x, double* misfit,
                              double* grad, double* dsyn, int32_t* flag);
/* Same with HOST pointers (copies in and out, synchronises). */
int rfs_joint_misfit_grad(rfs_ctx* ctx, int nchain, const double* x, double* misfit, double* grad,
                          double* dsyn, int32_t* flag);
/* forward(x) of the plugins (model_rf_swd_vs_thk.py:27-49): synthetics only.
 * Note the reference quirk kept here: every SWD block is computed at tRc (model_surf.py:114-130)
 * when `quirk_trc_everywhere` != 0. */
int rfs_joint_forward(rfs_ctx* ctx, int nchain, const double* x, int quirk_trc_everywhere,
                      double* dsyn, int32_t* flag);

/* -------- caller of the path: leapfrog trajectory (pyhmc/hmc.py:121-201) --------------- */
/* One device-resident trajectory for nchain chains: given x0, p0 (DEVICE, [nchain][2n], p0 already
 * drawn), per-chain dt and L, and bounds[2n][2], run the reference's leapfrog with mirror
 * reflection (hmc.py:121-137, 164-183; hmcda.py:246-271).  Outputs (DEVICE): xnew, Ucur, Unew,
 * Hcur, Hnew, dsyn_cur, dsyn_new [nchain][ndata], ok[nchain] (0 where the reference would return
 * early: flag False or NaN in x / grad / dsyn).  Every output is written for every chain: a chain with ok = 0
 * gets xnew = x0 and Hnew = +inf (the reference's (xcur, inf, dobs, False)).  1 <= L[chain] <= Lmax is required;
 * a chain that violates it is reported with ok = 0. */
int rfs_leapfrog_dev(rfs_ctx* ctx, int nchain, const double* x0, const double* p0, const double* dt,
                     const int32_t* L, int32_t Lmax /* max over chains of L, known to the host that drew L */,
                     const double* bounds, double* xnew, double* Ucur,
                     double* Unew, double* Hcur, double* Hnew, double* dsyn_cur, double* dsyn_new,
                     int32_t* ok);

/* Same, for chains SORTED BY DECREASING L: nactive[step] (HOST array [Lmax], non-increasing) = number of chains with
 * L > step; step `step` then evaluates only the first nactive[step] chains instead of all of them.  With per-chain
 * trajectory lengths (HMC draws L per chain, pyhmc/hmc.py:248; dual averaging derives it from the per-chain dt,
 * pyhmc/hmcda.py:307) this removes the evaluations of chains that already finished.  nactive == NULL: as above. */
int rfs_leapfrog_dev2(rfs_ctx* ctx, int nchain, const double* x0, const double* p0, const double* dt,
                      const int32_t* L, int32_t Lmax, const int32_t* nactive, const double* bounds, double* xnew,
                      double* Ucur, double* Unew, double* Hcur, double* Hnew, double* dsyn_cur, double* dsyn_new,
                      int32_t* ok);

/* Continuous-flow variant: every chain is at its own point of its own trajectory, so chains with short trajectories
 * never wait for the longest one.  All arrays DEVICE, persistent between calls and owned by the caller:
 *   x, p [nchain][2n]; dt [nchain]; rem [nchain] leapfrog steps still to do (-1 = idle);
 *   fresh [nchain] = 1: the trajectory starts with this call (x = start model, p = drawn momentum, rem = L, ok = 1).
 * One context advances one such state at a time: the warm start of the root search (roots, kernels and model of the
 * previous evaluation) belongs to the state whose x array the previous call was given -- a call with another x array
 * starts over from the reference-semantics search (correct, but nothing is continued), and with "flow_async_handback" on
 * it is refused while searches of the other state are outstanding.  Two states that advance in turn want two contexts.
 * One call = ONE misfit+gradient evaluation of every chain: a fresh chain gets Ucur, Hcur, dsyn_cur and the first half
 * kick (pyhmc/hmc.py:150-164); a running chain drifts (with mirror reflection), is evaluated and kicked, rem--; when
 * rem reaches 0 its Unew, Hnew, dsyn_new are final, x holds the end model and done[chain] = 1.  ok[chain] = 0 (and
 * done = 1) where the reference returns early (flag False or NaN).  The caller restarts finished chains (accept /
 * reject, new momentum, rem = L, fresh = 1) between calls. */
int rfs_flow_step(rfs_ctx* ctx, int nchain, double* x, double* p, const double* dt, int32_t* rem, int32_t* fresh,
                  const double* bounds, double* Ucur, double* Hcur, double* Unew, double* Hnew, double* dsyn_cur,
                  double* dsyn_new, int32_t* ok, int32_t* done);

/* rfs_flow_step with restarts on the device.  The reference's chain, after a trajectory (pyhmc/hmc.py:192-198, 246-258):
 * draws u ~ U(0,1), accepts the end point if u < exp(-(Hnew - Hcur)), draws the next L and momentum, starts over.  None of
 * those draws depends on the trajectory, so a caller can make them AHEAD of time and deposit them here (have[chain] = 1
 * before the call in which the chain's rem reaches 0).  The device then does the accept / reject itself -- x[chain] stays at
 * the end model or goes back to xstart[chain], the model the trajectory started from, which the calls maintain -- takes the
 * next momentum and length, sets fresh = 1 and reports done[chain] = 2 (rejected) or 3 (accepted): the chain evaluates its
 * new start model in the very next call instead of sitting one out while the host catches up.  What the host still needs
 * for its books is parked where the next trajectory does not touch it: res_x (end model), res_val = {Ucur, Hcur, Hnew, Unew},
 * res_dsyn (synthetics at the end model; may be NULL).  have[chain] is cleared when consumed.  A chain without a deposit,
 * or one that fails (ok = 0), behaves exactly as in rfs_flow_step (done = 1, the host restarts it) -- and keeps its deposit,
 * which the host has to withdraw (have = 0).  The threshold exp(-(Hnew - Hcur)) is evaluated with the device's exp: against
 * a host that evaluates it with its own libm the decision can differ only where u lies within one ulp of it.
 * next == NULL: rfs_flow_step.  The struct lives on the HOST, every pointer in
 * it is a DEVICE pointer.
 * Deferred form (rem == NULL, gsave and kick given) for samplers whose next step size depends on the trajectory just
 * completed (dual averaging, pyhmc/hmcda.py:329-345): the deposit holds only u and p; a restarted chain gets a placeholder
 * length, and NO chain takes its first half kick p -= dt/2 grad in the call that evaluates its start model -- the
 * gradient is kept in gsave, kick[chain] = 1, and the kick is applied (same arithmetic) at the start of the next call.
 * The host therefore has one whole call to write the chain's new dt and rem.  gsave / kick may also be given together
 * with rem; all start models are then treated that way. */
typedef struct rfs_flow_next {
    int32_t* have;        /* [nchain] */
    const double* u;      /* [nchain] acceptance draw */
    const double* p;      /* [nchain][2*nlayer] momentum of the next trajectory */
    const int32_t* rem;   /* [nchain] its number of leapfrog steps */
    double* xstart;       /* [nchain][2*nlayer] */
    double* res_x;        /* [nchain][2*nlayer] */
    double* res_val;      /* [nchain][4] */
    double* res_dsyn;     /* [nchain][ndata] or NULL */
    double* gsave;        /* [nchain][2*nlayer] or NULL, see below */
    int32_t* kick;        /* [nchain] or NULL */
} rfs_flow_next;
int rfs_flow_step2(rfs_ctx* ctx, int nchain, double* x, double* p, const double* dt, int32_t* rem, int32_t* fresh,
                   const double* bounds, double* Ucur, double* Hcur, double* Unew, double* Hnew, double* dsyn_cur,
                   double* dsyn_new, int32_t* ok, int32_t* done, const rfs_flow_next* next);

/* rfs_flow_step2 that also hands the caller's books what they need of every chain that COMPLETED a trajectory in the call, packed
 * densely into memory the host can read -- instead of the caller copying the done flags down, building index lists, copying them
 * up and gathering rows (pyhmc's loop: a dozen small copies and index kernels per step; with this, none).
 *   records->buf: memory the DEVICE can write and the HOST can read: pinned host memory (hipHostMalloc / a torch tensor with
 *   pin_memory=True: mapped into the device's address space at the same address), zero-filled by the caller, used as a RING of
 *   `cap` records (cap >= 2 nchain) of stride 8 + 2 nlayer (+ ndata with want_dsyn) float64 each:
 *     [0] chain, [1] done code (1: waits for the caller, as in rfs_flow_step; 2 / 3: rejected / accepted and restarted on the
 *     device from its deposit), [2] ok, [3] Ucur, [4] Hcur, [5] Hnew, [6] Unew, [7] `stamp` of the call that wrote the record,
 *     [8 ..) the end model, [8 + 2 nlayer ..) its synthetics.  For done codes 2 / 3 these are the values rfs_flow_next parks in
 *     res_val / res_x / res_dsyn.
 *   Record number k since the last `reset` lies in slot k % cap.  The caller gives every call its own non-zero stamp, keeps a
 *   read cursor and -- behind an event recorded after the call -- takes records while their [7] equals that call's stamp (what
 *   follows is a stale record of an earlier lap or one of the NEXT call, which may already be running: cap >= 2 nchain keeps
 *   the call being read and the one under way apart).  The order of a call's records is not defined (sort by chain).
 *   reset != 0: the ring starts over at slot 0 with this call (first call of a run).  records == NULL: rfs_flow_step2. */
typedef struct rfs_flow_records {
    void* buf;
    uint64_t bytes;     /* size of buf: >= 8 * cap * stride */
    int32_t cap;        /* slots of the ring */
    int32_t want_dsyn;  /* != 0: every record carries the ndata synthetics as well */
    double stamp;       /* non-zero, different from the stamps of the two calls before */
    int32_t reset;
} rfs_flow_records;
int rfs_flow_step3(rfs_ctx* ctx, int nchain, double* x, double* p, const double* dt, int32_t* rem, int32_t* fresh,
                   const double* bounds, double* Ucur, double* Hcur, double* Unew, double* Hnew, double* dsyn_cur,
                   double* dsyn_new, int32_t* ok, int32_t* done, const rfs_flow_next* next, const rfs_flow_records* records);
/* The deposits of rfs_flow_next for n chains in ONE launch on `hip_stream` (NULL: the context's stream): for i < n, chain idx[i]
 * gets u[i], pnew[i][2 nlayer], remnew[i] (NULL in the deferred form) and have = 1 -- what pyhmc/hmc.py:246-258 draws for the
 * trajectory after the one that is running.  idx / u / pnew / remnew: device memory, or pinned host memory mapped into the device
 * (a few KB read over the link cost less than a copy in front of the launch; above ~64 KB copy first).  The caller orders the
 * launch before the call in which the chains complete (an event on hip_stream that the calls' stream waits for). */
int rfs_flow_deposit(rfs_ctx* ctx, void* hip_stream, int nchain, int n, const int32_t* idx, const double* u, const double* pnew,
                     const int32_t* remnew, const rfs_flow_next* next);

/* What the caller does to the chains that go through the host between two flow steps -- pyhmc/hmc.py:228-276's loop body
 * for a chain whose trajectory ended in a failed evaluation (no acceptance draw: the reference returns early, :154-180), or
 * every finished chain of a run without rfs_flow_next -- in one launch on the context's stream instead of a dozen small
 * copies and scatters between two steps.  All pointers DEVICE: idx1 [n1] chains whose row of x becomes xkeep [n1][2*nlayer]
 * (the model the chain keeps); idx2 [n2] chains that start another trajectory: p <- pnew [n2][2*nlayer], rem <- remnew [n2],
 * (DEVICE, or pinned host memory mapped into the device: the launch then reads the lists over the link -- right for a few KB,
 * above ~64 KB copy them to the device first)
 * dt <- dtnew [n2] (NULL: unchanged), fresh = ok = 1 -- or, with pnew NULL, chains the device has already restarted whose
 * length and step size follow a call late (rfs_flow_next.rem == NULL, pyhmc/hmcda.py:280-369): rem and dt only;
 * idx3 [n3] chains whose deposit (rfs_flow_next.have) is withdrawn.
 * x, p, rem, dt, fresh, ok: the arrays of rfs_flow_step; nxt_have: rfs_flow_next.have or NULL (n3 = 0). */
int rfs_flow_restart(rfs_ctx* ctx, int nchain, int n1, const int32_t* idx1, const double* xkeep, int n2, const int32_t* idx2,
                     const double* pnew, const int32_t* remnew, const double* dtnew, int n3, const int32_t* idx3,
                     double* x, double* p, int32_t* rem, double* dt, int32_t* fresh, int32_t* ok, int32_t* nxt_have);

/* Diagonal inverse mass matrix of the leapfrog entries above (rfs_leapfrog_dev / dev2, rfs_flow_step): drift
 * x += dt * minv * p, kinetic energy p.minv.p / 2; the caller draws p ~ N(0, M).  minv: HOST [2*nlayer], NULL =
 * identity (the reference's `invert_Mass`, pyhmc/hmc.py:48).  Reset by rfs_joint_setup. */
int rfs_set_inverse_mass(rfs_ctx* ctx, const double* minv);

/* -------- introspection ---------------------------------------------------------------- */
int rfs_ndata(const rfs_ctx* ctx);      /* nt + ntRc + ntRg + ntLc + ntLg of the current joint setup */
/* Tuning knobs (no effect on results beyond last-bit rounding):
 *   "swd_lanes_per_chain"  lanes that share one chain's root search: 0 = automatic (<= 3072 (sequence, chain) items: the
 *                          latency form of the lanes-per-item kernel -- 64 / 32 / 16 lanes per item, see the next two
 *                          options; above: cooperative producer / consumer blocks), else a power of two <= 64
 *                          (1 = the sequential lane-per-chain kernel).
 *   "swd_segments"         lanes-per-item search only: the sequential vector recurrence of one secular evaluation is cut
 *                          into 1, 2 or 4 segments that run side by side on the item's lanes (rows of the segments'
 *                          products, folded afterwards); -1 (default) = 4 where latency is what counts (few items).
 *                          Same product, different association: roots agree to rounding.
 *   "swd_speculate"        lanes-per-item search only: 1, 2 or 4 wavefronts per block evaluate the next points of a
 *                          scan ahead of time; the state machine still consumes them one by one and only where it asks
 *                          for exactly that point, so results are bit-identical.  -1 (default) = automatic.
 *   "share_rc_rg"          1 (default): when the Rg block has exactly the Rc block's periods, its pass at T (sregnpu's
 *                          central pass) reads the Rc block's roots and eigenfunctions instead of computing them again
 *                          (Lg / Lc likewise);
 *                          0 = always compute it (set before rfs_joint_setup; results are bit-identical).
 *   "cu_split"             0 = RF kernels on the caller's stream, sharing CUs with the root search;
 *                          1 (default) / 2 = when the cooperative root search fits on half of the CUs, it and
 *                          the RF kernels run on disjoint halves of the CU mask (contiguous halves / even-odd
 *                          bits); the combine uses the whole chip.
 *   "rf_scratch_budget_mb" the fused gradient keeps one row vector per (chain, layer, frequency) between its two RF
 *                          sweeps (4.1 GB at 8192 chains x 30 layers x 512 samples); batches whose scratch would exceed
 *                          this budget (default 4096 MB) go through the RF pipeline in chain tiles.  Results are
 *                          bit-identical for every tile size.
 *   "recalibrate"          forget the measured schedule (the partition / early-eigenfunction decisions are taken from
 *                          HIP-event timings of the first evaluation of each shape and cached in the context).
 *   "rf_band_limit_digits" band limit of the fused frequency-domain RF gradient (rfs_joint_misfit_grad*, leapfrog / flow
 *                          entries).  Every frequency's contribution to the gradient carries the Gaussian
 *                          G = exp(-(w / 2 f0)^2) (RFModule.f90:393) over a denominator >= water * max (:413-419); where
 *                          G < 10^-digits * water it cannot reach the sum in double precision.  The row sweep still
 *                          visits those frequencies (the water level is a maximum over all of them, :396-398, and the
 *                          forward trace uses the whole spectrum) but stores no rows for them, and the adjoint sweep
 *                          skips them: 42 % of the frequencies at nt = 512, dt = 0.1, f0 = 1.5, 85 % at nt = 2048,
 *                          dt = 0.025.  Default 13 (gradients agree with the unlimited sum to ~1e-13); 0 = no limit.
 *                          librf's kernel_all (rfs_rf_kernel_all) and the time-domain method are never limited.
 *   "rf_band_floor_digits" a lane of the RF sweeps is a frequency and a wavefront 64 of them: where a multiple of 64 bins lies
 *                          between the limits of "rf_band_floor_digits" (default 8) and "rf_band_limit_digits" the band
 *                          ends there, so that no wavefront runs mostly idle (nt = 512, dt = 0.1, f0 = 1.5: 128 bins instead
 *                          of 150; measured difference to the unlimited gradient 8e-16).  0 = never move the limit down.
 *                          Worst case: a dropped frequency carries at most 10^-floor x the weight of the strongest one, so
 *                          the gradient's RELATIVE error is bounded by (dropped bins) x 10^-floor -- reached only where the
 *                          water level clamps the spectrum beyond the band (a water level of order 0.1 with a small f0); the
 *                          8e-16 / 1e-13 figures are those of spectra the water level does not clamp.  The same floor
 *                          bounds the weight of the float32-swept bins in the forward trace ("rf_f32_beyond_band").
 *   "swd_warm_start"       the root search inside a trajectory.  The reference searches every model from scratch, period
 *                          after period (surfdisp96.f:257-316: ~23 secular evaluations per period, each period starting
 *                          from the root before it).  Inside a leapfrog trajectory the model of step s is the model of
 *                          step s-1 moved by dt M^-1 p, and step s-1 left its roots AND their Frechet kernels on the device:
 *                          every (period, chain) item predicts its root to first order, brackets it inside a trust radius
 *                          and refines it by false position (~3 secular evaluations, all periods in parallel).  Chains
 *                          that cannot be continued -- first evaluation, a failed previous evaluation, no sign change
 *                          where the first-order model says, root above the fastest layer -- go through the
 *                          reference-semantics search, which alone decides flags.  Accepted roots are sign changes of the
 *                          very secular function the reference brackets, located to 1e-7 c and rounded to float32 like
 *                          the reference's: within the reference's own refinement tolerance 1e-6 c (surfdisp96.f:627) of
 *                          its values, not bit-identical to them (its nevill ends in bisection steps and stops up to
 *                          1e-6 c short of the root).
 *                          0 = off: every evaluation by the reference-semantics search (bit-exact roots);
 *                          1 (default) = on for the trajectory entries (rfs_leapfrog_dev / _dev2 without nactive,
 *                          rfs_flow_step / _step2), whose consecutive evaluations are one set of chains moving;
 *                          2 = also for rfs_joint_misfit_grad[_dev]: the CALLER promises that consecutive calls with the
 *                          same nchain evaluate the same chains a small step apart (a host-side leapfrog loop).
 *                          rfs_swd_forward / rfs_swd_kernel / rfs_joint_forward never warm-start.
 *   "swd_warm_exact"       1 (default): behind the warm start, every continued root is turned into THE REFERENCE'S root: the
 *                          reference's nevill returns the last midpoint of a run of bisections, 0.5 .. 1e-6 c short of the sign
 *                          change, a function of the scan cell the bracket was found in -- so each lane takes a group of
 *                          consecutive periods of one sequence, rebuilds the reference's scan grid period by period (origin =
 *                          the unrounded root of the period before - 1.5 dc, surfdisp96.f:272-275), steps to the cell that holds
 *                          the continued root and runs the reference's own refinement (surfdisp96.f:568-687) inside it.  The
 *                          hand-over between periods is sequential only inside a group; a group starts with
 *                          "swd_exact_runup" extra periods whose only purpose is its first origin (an origin that is off
 *                          by 1e-6 c moves the result by ~1e-9 c).  Measured (tests/
 *                          test_gpu_warm.py, 6.9 M roots): >= 99.99 % of the roots bit-identical to the reference-semantics
 *                          search, the rest within 1e-6 c; misfit and gradient as with "swd_warm_start" = 0.
 *                          0: keep the converged roots (within 1.1e-6 c of the reference's, misfits to ~1.4e-5, gradients to
 *                          ~1e-5 except on ill-conditioned chains): 2-3 times cheaper in the root search.
 *   "swd_exact_group" / "swd_exact_runup"   periods per lane (default 4, >= 2 -- 1 only with 16 lanes per group, "swd_exact_coop": measured best at 8192 chains x 40 periods -- fewer make more lanes and more run-up work, more make the kernel's dependent chain longer) and run-up periods (default 2; 1 with "swd_exact_origin_tol_e9" 500 is the faster, looser setting described there) of the above.
 *   "swd_walk_window"      2 (default): which periods of a sequence with anomalous dispersion walk the reference's scan grid for the
 *                          branch test.  -1 = all of them (rounds 3-5: 540 evaluations per such sequence and step, 2.0 M of
 *                          the step's 6.3 M -- the first period alone scans 100-700 cells from the model's start value); W >= 0 =
 *                          the periods within W of an anomalous pair (c(j) <= c(j-1) - 1.5 dc: the pair itself, W periods
 *                          before and after); the others take the test of a sequence with normal dispersion -- one evaluation
 *                          at the point their scan starts from.  A chain marked wide ("swd_warm_widen") and a sequence that
 *                          walks for a root next to the fastest layer still walk whole.  Measured at 8192 chains: 9.6 -> 6.7
 *                          evaluations per item in the warm stage, 4.75 -> 4.54 ms per step (same box; W = 0 / 1 / 4: 4.65 / 4.63 /
 *                          4.56), 29 -> 27 chains per step handed back.  Against the oracle (the sampler run stopped at eighteen
 *                          device steps, 108 238 mid-trajectory chains + 10 702 end models with W = 2): every root within the
 *                          reference's own bracket of the oracle's (<= 1.8e-6 c), 94.3 % of the chains with all 40 roots
 *                          bit-identical -- the figures of -1 (profiles/r06_flow_parity_stats*.txt).
 *   "swd_exact_budget"     44 (default): the reference-root stage of big batches runs in two launches.  Its wavefronts execute what
 *                          their slowest lane needs (47 evaluations where the lanes need 38.5 on average: the lazy nevill's
 *                          count varies from period to period), and the stage ends with its slowest wavefront.  Every lane gets
 *                          this many evaluations; the groups that are unfinished then (6 % at 44) have their machines saved --
 *                          in the middle of a period if need be -- and are continued by k_swd_exact_coop, 16 lanes per group.
 *                          The same evaluations in the same order: the same roots bit for bit (test_gpu_warm.py).  4.71 -> 4.64 ms
 *                          per step at 8192 chains (40: 4.69, 48: 4.68, 36: 5.10 -- too many groups passed on).  0: one launch.
 *   "swd_exact_overlap"    1: in the background form of the flow entries the second launch of the above runs on the walk stream BESIDE
 *                          the eigenfunction pass of all items, and the periods of the groups it finishes get their
 *                          eigenfunctions again afterwards (k_swd_eigen_groups).  Same results; measured slower (4.55 -> 4.67 ms:
 *                          the launch starves beside the pass and the RF sweeps).  0 (default): one after the other.
 *   "swd_exact_redo_runup" r > "swd_exact_runup": a group whose run-up did not bring its first origin within the tolerance is not
 *                          handed to the sequential search but done again with r run-up periods (16 lanes per group); only a
 *                          second failure hands the chain back.  -1 (default) = 4 for the small batches' 16-lane form (ONE
 *                          configs[0] chain: 59 of 399 evaluations handed back for this cause, none with the second try), 0 for
 *                          big batches (0.1 chains per step there).  Parity: more run-up is never looser.  NOT a way to one
 *                          run-up period: with "swd_exact_runup" 1 + r = 3 the stage does a sixth less work, but a soak with
 *                          group velocities still shows 73 of 518 473 roots a float32 step off (1 with two run-up periods) -- the
 *                          ordinary groups' 4e-10 c of origin error is enough.  0 = never.
 *   "rf_store_hyp"         1: with row peeling, pass A leaves exp / cos / sin of every (layer, band frequency) in HBM for pass B
 *                          (1.5 GB at 8192 chains): 10 % fewer instructions in pass B, 2.9 GB more traffic per step -- measured
 *                          3 % SLOWER.  0 (default).
 *   "swd_exact_coop"       1 (default): batches of up to 8192 (group, chain) pairs run the reference-root stage with 16 lanes per
 *                          group -- each lane builds the layer entries of every 16th layer, every lane runs the short vector
 *                          recurrence: the single lane's arithmetic operation for operation, the same roots bit for bit, a third
 *                          of the time per evaluation: a stage as long as one lane's ~40 dependent evaluations whatever the
 *                          batch size is what a small batch spends most of its step in.  0 = a lane per
 *                          group always, 2 = 16 lanes per group always (tests).  Models of up to 65 layers.
 *   "swd_cold_scan"        -1 (default): in batches of up to 64 chains whose hand-backs are searched in the foreground (not
 *                          "flow_async_handback"), a chain the warm start itself declines -- no previous evaluation, a move the
 *                          first-order model cannot follow, no sign change inside the trust radius -- takes the search WITHOUT
 *                          a prediction instead of the sequential one: every period's secular function on one grid (start value
 *                          + i dc, lane = grid point), each sign change refined to 1e-6 km/s, the reference's scan replayed on
 *                          those roots period after period (k_swd_cold_scan, k_swd_cold_pick) -- then the branch test on the
 *                          reference's own grid for every period and the reference-root stage, as for any continued root; what
 *                          they decline goes to the sequential search after all.  ~0.1 ms where the sequential search of ONE
 *                          chain takes 2 ms.  Such batches also keep ONE hand-back list and ONE sequential search per step, on
 *                          the step's own stream (12 launches on the step's chain instead of 19).  0 = off, 1 = batches of up
 *                          to 512 chains.  Sequences of up to 192 periods, fundamental mode, no water layer.
 *   "swd_cold_again"       1 (default): in those batches (beyond "swd_cold_first") a chain whose last evaluation ended on the
 *                          hand-back list goes straight to the search without a prediction the next time -- a wild chain is wild
 *                          for many steps, and the branch test would decline its continued roots again (32 wild chains: 64 -> 48
 *                          sequential searches in 200 steps, -5 % per step).  Kept per chain across the steps the chain sits
 *                          out: no result depends on the host's timing.  0 = off.
 *   "swd_exact_group_small"  2 (default): periods per group of the reference-root stage in those small batches (16 lanes per
 *                          group; the stage is as long as a group's periods + run-up one after the other, and the groups whose
 *                          run-up does not contract get a second try behind eight run-up periods there).
 *   "swd_cold_first"       8 (default): batches of up to that many chains skip the warm search and give every chain the search
 *                          without a prediction (configs[0]: one chain per rank) -- a chain the branch test declines after a
 *                          continued root would otherwise cost a sequential search (8 wild chains: 0.99 -> 0.91 ms per step;
 *                          16: the same either way; 32: 1.03 -> 1.24).  0..512.
 *   "flow_async_handback"  rfs_flow_step / rfs_flow_step2 with the warm start on: 1 = a chain the warm start hands back to the
 *                          reference-semantics search (a few per step on rough models: ~3 ms of dependent evaluations, during
 *                          which every other chain would wait) sits that step out instead -- its model has drifted, it is not
 *                          evaluated, kicked or counted (statistic flow_chain_steps); its search runs on a side stream beside
 *                          the NEXT call, which completes the chain's step from those roots (no second drift).  Each chain
 *                          still goes through exactly the same sequence of models, evaluations and decisions; only the
 *                          call in which a given leapfrog step of a given chain happens moves by one.  A caller that
 *                          counts calls (rem steps = rem calls) must look at rem / done instead.  0 (default) = every call
 *                          completes every chain's step.  The samplers of pyhmc switch it on for sample_flow.
 *   "swd_exact_origin_tol_e9"  how far (in units of 1e-9 c, default 100) the origin of a wanted period's scan grid may be from the
 *                          reference's, as tracked through the run-up periods, before its group is handed back.  ONE run-up period
 *                          ("swd_exact_runup" 1) needs 500 here: a group whose run-up period was closed by bisections alone keeps an
 *                          origin up to 4e-7 c off (0.3 % of the groups; with 100 they would all be handed back).  That setting does
 *                          a sixth of the stage's evaluations less (-3.7 % per step, same-box A/B) and, against the oracle on 3 072
 *                          completed + 3 072 mid-trajectory chains of the burned-in bench population (phase velocities only), gives
 *                          the default's figures (misfit max 2.9e-6 / 3.6e-6, none above 1e-5; gradient 6 / 4 chains above 1e-5) --
 *                          but on random configurations with group velocities, which difference the roots of neighbouring periods,
 *                          79 instead of 1 of 518 473 roots differ from the sequential search's by a float32 step and the misfit
 *                          is off by up to 5.1e-5 instead of 6.0e-6.  Hence an option, not the default.
 *   "swd_warm_widen"       1 (default): a warm search that finds no sign change within its trust radius (the root has left the
 *                          first-order model's reach: 32 chains per step of a burned-in 8192-chain population) keeps widening its
 *                          bracket, out to 16 x the radius or 0.1 km/s; an item whose first-order change exceeds 2 km/s (kernels
 *                          blown up next to an osculation point) looks around its previous root the same way.  Whatever root is
 *                          found there is not taken on the first-order model's word: every sequence of such a chain walks the
 *                          reference's scan grid, and the walk decides (-> the full search otherwise).  0: such chains go to the
 *                          full search at once (round 4's behaviour: 54 instead of 39 chains per step handed back).
 *   "swd_warm_feedback"    1 (default): error feedback of the warm start's predictor.  What the first-order prediction missed by at
 *                          the previous step (root - (previous root + G . dx): the second-order term of the root along the
 *                          trajectory) is added to this step's prediction -- consecutive moves of a trajectory are nearly
 *                          equal, and so are their second-order terms; nothing is carried into the first step of a trajectory.
 *                          A search that fails from the corrected prediction starts over from the plain one.  No result
 *                          changes (the roots are the same sign changes); 0.6 evaluations per item less and a third fewer
 *                          hand-backs (39 -> 29 chains per step at 8192 chains: the second attempt also rescues searches that
 *                          failed before).  0 = off.
 *   "swd_warm_round_budgets"  303 (default) = b1 + 100 b2 + 10000 b3: the warm search runs in ROUNDS.  A wavefront executes what its
 *                          slowest lane needs, and the searches are very uneven (2 evaluations where the Newton start brackets
 *                          the root at once, 3-6 through a bracket, 20-60 through a widened bracket and bisection: 2.56
 *                          evaluations per lane, 7.98 executed per lane in one round).  Every lane gets b1 evaluations; the
 *                          searches that are unfinished then are packed densely into a list and get b2 more in a second launch
 *                          (b3: a third), and a last round without a limit finishes the rest.  A lane's sequence of
 *                          evaluations does not depend on who shares its wavefront: results are bit for bit those of one
 *                          round (0).  5.06 -> 4.95 ms per step at 8192 chains with (2, 3); with the cooperative last round
 *                          and (3, 3): 4.62.
 *   "swd_warm_last_round_coop"  1 (default): the last round -- a few per cent of the items, and the stage ends with its slowest
 *                          search -- with 16 lanes per search: each lane builds the layer entries of every 16th layer, every
 *                          lane runs the short vector recurrence (the arithmetic of the single lane's evaluation, operation for
 *                          operation: same roots bit for bit, a third of the time per evaluation).  Models of up to 65 layers;
 *                          0 or deeper models: one lane per search.
 *   "flow_skip_idle"       1 (default): in the flow entries a chain that is idle in a step -- waiting for the caller after a
 *                          trajectory, or failed -- is neither continued nor handed back (nothing reads its evaluation, and a
 *                          failed chain would go to the full search at every step it waits).  0: round 4's behaviour.
 *   "swd_warm_reset"       (any value) forget the previous evaluation: the next one goes through the reference-semantics
 *                          search for every chain.  The samplers of pyhmc call it whenever they write a checkpoint, so
 *                          that a resumed run and the uninterrupted one evaluate the same way from there on (their own
 *                          schedules start every trajectory batch / flow segment from the full search anyway; a caller's
 *                          host-side leapfrog loop with swd_warm_start = 2 needs the call).
 *   "swd_warm_serial"      1: a warm-started step runs on ONE stream (every kernel alone on the chip: clean per-kernel
 *                          durations for profiling); 0 (default): the surface-wave kernels on a second stream beside the
 *                          receiver-function sweeps (~5 % faster).  Results are identical.
 *   "rf_row_peeling"       frequency-domain RF adjoint of the joint entries: the column sweep (pass B) obtains the row of
 *                          layer j from the row of layer j-1 times A_j^-1 (the propagator over -h), starting from the row
 *                          sweep's final row, instead of reading one stored row per (layer, frequency) -- for every chain whose
 *                          layer matrices stay close to unitary: growth exponent sum_j h_j (sigma |Im p_beta| + w_max |Re p_beta|)
 *                          <= 5, i.e. a teleseismic slowness and a time window (sigma = 4 / window) not much shorter than the
 *                          S travel time through the stack; decided per chain on the device, the other chains keep their
 *                          stored rows.  Rows agree with the stored ones to 1e-14 over 50 layers at the bench's window.
 *                          -1 (default) / 1 = as described, 0 = stored rows for every chain, 2 = peel every chain (diagnostics).
 *   "rf_peel_check"        1: the column sweep also checks the peeling's closure -- with every layer taken off, the row must
 *                          be the half-space's own -- and keeps the largest relative miss (rfs_get_stat "rf_peel_residual",
 *                          in units of 1e-18: ~1e4 = 1e-14 where the waves propagate).  0 (default) = off.
 *   "rf_f32_beyond_band"   frequency-domain RF with a band limit (rf_band_limit_digits): the frequencies beyond the band reach
 *                          the results only through the water level (a maximum of |R21|^2 over ALL frequencies,
 *                          RFModule.f90:396-398) and through spectrum values weighted by exp(-(w/2f0)^2) < 1e-11.  1 (default):
 *                          the row sweep takes them in float32 for every chain whose layer matrices stay tame up to the
 *                          Nyquist frequency (n e^{2E} <= 1e4, E the growth exponent above), then decides per chain from the
 *                          EXACT band values whether more is needed: not if the band holds both maxima, or if no band
 *                          frequency can reach a water level set by 1.01 x the float32 maxima; otherwise that chain's
 *                          frequencies are swept again in f64 (rfs_get_stat "rf_f32_resweeps").  Trace, misfit and gradient
 *                          equal the all-f64 sweep to a few 1e-13.  0 = f64 for every frequency.
 *   "rf_mid_fused"         1 (default): the middle section of the frequency-domain gradient -- water level, spectrum, inverse FFT,
 *                          trace, residual, misfit, forward FFT of the weighted residual -- runs as ONE kernel with a chain's 4 KB
 *                          in LDS (FFT lengths 16 .. 4096) instead of two kernels around two rocFFT calls; same numbers to
 *                          rounding (1e-15).  librf's entries, the forward-only calls and the time domain always use rocFFT.
 *   "swd_walk_dense"       1 (default): the grid walk of a walking sequence's later periods evaluates exactly the points the
 *                          reference's scan passes on its way to the continued root (their number follows from the root),
 *                          dealt densely to the lanes; 0: 8 lanes per item evaluate rounds of 8 grid points speculatively.
 *                          Same verdicts, about half the evaluations.
 *   "swd_exact_final"      1: with the warm start on, the start model and the end model of every trajectory (the two
 *                          evaluations the accept / reject decision and the stored sample come from) still go through the
 *                          reference-semantics search.  0 (default) = off.
 *   "early_eigen_periods"  in a partitioned step the RF half finishes before the root search; the eigenfunction
 *                          kernels of the first periods (whose roots are final by then) run there early and only
 *                          the rest waits for the search.  -1 (default) = automatic count, 0 = off, k > 0 = the
 *                          first k periods.  Results are bit-identical for every value. */
int rfs_set_option(rfs_ctx* ctx, const char* name, int value);
/* Counters (cumulative since the context's warm-start buffers were made; the call synchronises):
 *   "flow_chain_steps"          (chain, step) pairs by which rfs_flow_step / rfs_flow_step2 advanced a trajectory (idle chains
 *                               -- waiting for the host, or failed -- do not count): the evaluations a sampler used
 *   "rf_f32_chains"             chain evaluations whose frequencies beyond the band were swept in float32; "rf_f32_resweeps":
 *                               those of them swept again in f64 (see "rf_f32_beyond_band")
 *   "rf_band_bins" / "rf_bins"  frequencies inside the band of the fused gradient ("rf_band_limit_digits"; the rest is what
 *                               "rf_f32_beyond_band" sweeps in float32) / all frequencies n2 of the joint configuration
 *   "swd_warm_declined_chains"  chain evaluations the warm start handed back to the reference-semantics search
 *   "swd_warm_items"            (period, chain) items the warm start refined
 *   "swd_warm_secular_evals"    secular-function evaluations it spent on all items
 *   "swd_warm_walked_chains"    chain evaluations whose sequences walked the reference's scan grid (anomalous dispersion, or
 *                               a first-order change above 0.5 km/s: "swd_warm_wide_chains" counts the latter)
 *   "swd_exact_secular_evals"   secular-function evaluations of the reference-root stage ("swd_warm_exact");
 *   "swd_exact_evals_slowest_lane" / "swd_exact_wavefronts"   divergence of that stage (k_swd_exact): evaluations of each
 *                               wavefront's slowest lane, summed (a wavefront executes 64 x that), and wavefronts run
 *   "swd_exact_declined_chains" chain evaluations it handed back (no sign change in the expected scan cell, a grid at the
 *                               floor of the scan, a root the reference rejects)
 *   "swd_exact_cause_<k>"       ... by cause: 1 no usable approximate root / origin, 2 root too far from the origin, 3 the grid touches
 *                               the floor of the scan, 4 no sign change in the root's cell nor in its neighbour, 5 root above the
 *                               fastest layer, 6 NaN, 7 the run-up left the origin short of "swd_exact_origin_tol_e9"
 *   "swd_warm_fail_no_change" / "swd_warm_fail_other"   (period, chain) items whose warm search failed: no sign change out to
 *                               the widest bracket / anything else
 *   "swd_warm_passed_on_<r>"    searches round r = 1, 2, 3 of the warm search passed on to the next round
 *   "swd_cold_chains" / "swd_cold_secular_evals"   chain evaluations that came through the search without a prediction
 *                          ("swd_cold_scan"; they are taken off "swd_warm_declined_chains", which then counts the sequential
 *                          searches) and its secular evaluations
 *   "swd_cold_fail_<c>"    ... and the chains it left on the list, by cause c = 34..39 (rfsurf_kernels.hpp, k_swd_cold_pick)
 *   "wstat_<i>"            slot i = 0..39 of the warm start's counter array as it is (what the names above read; diagnostics)
 *   "swd_warm_search_evals" / "swd_warm_search_evals_slowest_lane" / "swd_warm_search_lanes"   divergence of the warm search
 *                               (k_swd_warm): evaluations of all searches, of each wavefront's slowest search summed over
 *                               the rounds' wavefronts (what the wavefronts execute; a wavefront of the cooperative last round
 *                               holds 4 searches, the others 64), searches made.  One round: 2.56 evaluations per search needed,
 *                               40 400 wavefront-evaluations executed per step of 8192 chains; in rounds: 23 300
 *   "swd_warm_cause_<k>"        chains handed back, by (first) cause: 4 no usable previous evaluation / forced, 5 step too
 *                               large for a first-order model, 6 no sign change inside the trust radius, 7 root above the
 *                               fastest layer, 9 / 11 another root lies between the point the reference's scan of that
 *                               period (/ of a sequence's first period) starts from and the continued root, 8 / 10
 *                               degenerate start point */
int rfs_get_stat(rfs_ctx* ctx, const char* name, int64_t* value);
/* Diagnostics: the roots of the LAST evaluation of the joint configuration as they lie in the context's persistent root
 * buffer -- one value per (sequence, period) item in the order Rc, [Rg pass at T, 1.05 T, 0.95 T], Lc, ... (the float32
 * values the reference stores, surfdisp96.f:302; for a flow step: of every chain that was evaluated in it).
 * *nitems = items per chain (also with croots == NULL); croots: HOST [nchain][*nitems].  Synchronises. */
int rfs_last_roots(rfs_ctx* ctx, int nchain, int32_t* nitems, double* croots);
/* Kernel groups of one rfs_joint_misfit_grad_dev call.  With timing enabled every group of every
 * call is bracketed by its own pair of HIP events recorded on the stream the kernels run on
 * (the root search / eigenfunction groups run on the context's second stream); nothing
 * synchronises until rfs_kernel_ms_sum() sums the elapsed times (ms) and launch counts per
 * group and resets the accumulators.  rfs_enable_timing: on = 0 off, 1 every group, 2 * mask (mask bit = rfs_kernel_id)
 * only those groups -- every event pair costs a few microseconds of the step, so a caller that needs one group's
 * duration inside a timed region enables just that one. */
typedef enum {
    RFS_K_PREP = 0, RFS_K_RF_PASS_A, RFS_K_RF_MID, RFS_K_RF_PASS_B, RFS_K_SWD_ROOTS, RFS_K_SWD_EIGEN,
    RFS_K_COMBINE, RFS_K_SWD_EXACT /* the reference-root stage behind a warm start ("swd_warm_exact") */,
    RFS_K_FLOW_STEP /* one whole rfs_flow_step / rfs_flow_step2 call on the caller's stream: first launch .. behind the kick */, RFS_K_COUNT
} rfs_kernel_id;
int rfs_enable_timing(rfs_ctx* ctx, int on);
int rfs_kernel_ms_sum(rfs_ctx* ctx, double* ms /* [RFS_K_COUNT] */, int32_t* count /* [RFS_K_COUNT] */);
/* Where in a flow step each group runs, without a profiler attached: for every group the SUMS over the event pairs gathered
 * since timing was switched on of (group start - step start) and (group end - step start) in ms, step start = the first
 * launch of the rfs_flow_step2 call the pair belongs to (RFS_K_FLOW_STEP must be in the mask), and the number of pairs.
 * Synchronises; does not reset (rfs_kernel_ms_sum does).  Diagnostics: profiles/r05_step_timeline_events.txt. */
int rfs_kernel_timeline(rfs_ctx* ctx, double* start_ms /* [RFS_K_COUNT] */, double* end_ms /* [RFS_K_COUNT] */, int32_t* count /* [RFS_K_COUNT] */);

#ifdef __cplusplus
}
#endif
#endif /* RFSURF_H */
