#!/usr/bin/env python3
"""bench.py -- leapfrog-step throughput of the HIP hot path (BASELINE.json `metric`).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {1,3,4}] [--chains C] [--dt DT]

--gpus N launches N ranks BY ITSELF (one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set for
each, torch.distributed over RCCL); under `python -m torch.distributed.run ... bench.py --gpus N` (WORLD_SIZE
already in the environment) the process is one of those ranks.  Counterpart of the reference's
`mpiexec -n N python main_base.py` (main_base.py:16-18,59-60,90).  The launching process never touches the GPU.

Workloads (BASELINE.json `configs`; --config selects, configs[1] is the default and the headline):
  1  configs[1]: 8192 chains x 30-layer Vs+thk models per GPU, joint RF (P, nt = 512, dt = 0.1, Gaussian 1.5,
     shift 5 s, water 1e-3, freq method) + 40 Rayleigh phase periods linspace(5, 44, 40); plain HMC
     (HamitonianMC.sample_flow, L ~ U{5..20}) at a step size tuned for an acceptance ratio within 0.65-0.9
  4  configs[4]: the same with a 2048-point RF trace (dt = 0.025 s)
  3  configs[3]: HMCDualAveraging.sample_flow (main_DA.py), 8192 chains x 50 layers: per-chain dt and
     L = max(1, int(lambda / dt)), dual averaging included
Synthetic sorted-prior models, dobs = forward(true model).

A "step" = ONE DEVICE STEP OF THE SAMPLER: one leapfrog step of every chain through the C ABI (rfs_flow_step2: drift with
mirror reflection, misfit + gradient evaluation, kick; accept / reject and the next trajectory's start on the device from
draws made ahead; pyhmc/hmc.py:140-201, 228-276), state resident in HBM.  Set-up (untimed, like building a model): the chains
are burned in from their random start models for BURN_IN - W device steps; then W warm-up steps run untimed, then exactly K
steps are timed (W >= BURN_IN: the warm-up is the burn-in).  value = leapfrog steps (= misfit+gradient evaluations) of
chains inside a trajectory, all ranks / max-over-ranks wall time.
The JSON line of the default run (N = 1, configs[1]) also carries: the step size and the acceptance ratio of the timed
window, the root-search mode, `roofline` (the kernel group with the largest stand-alone time per step, its duration
measured live over the timed region), `valu_issue` (SQ_INSTS_VALU per step from the committed PMC passes), `root_search`
(evaluations per item, chains handed back), legs at other step sizes (`dt_sweep`: 0.002 / 0.02 / the reference's 0.1),
with the converged instead of the reference's roots (`converged_roots`), with the history-free search at every step
(`full_search_every_step`), rounds 1-3's headline definition (`never_ending_dt0002`), with every frequency of the RF row sweep
in f64 (`all_f64`), `config1_rg` (configs[1] + 40 group periods), `config4` / `config3` (dual averaging, timed in its
stationary regime: the adapting trajectories run untimed first) / `config0` (one chain), `roofline_fp64` (FP64 vector flops per
step from the committed counter passes against the measured step time), `host_cpu_ms_per_step` and `cpu_baseline`.  Independent chains shard across ranks (weak scaling, no data-path collective); the only collective is the
RCCL gather of the per-chain misfits after the timed region (comm.Gather, main_base.py:90).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

RAY_P, GAUSS, TSHIFT, WATER = 0.045, 1.5, 5.0, 0.001
NPER = 40
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s HBM3E
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X vector FP64: 256 CUs
METRIC = "leapfrog steps/sec (= forward+grad evals/sec) per GPU and whole node, 30-layer model"
# param.yaml:40 runs plain HMC at dt = 0.1; SURVEY 8(d): "tune so accept ~ 0.65-0.9".  Measured on the bench's chains after
# burn-in (scripts/dt_sweep.py, 8192 chains, L ~ U{5..20}): 0.998 at 0.002, 0.99 at 0.02, 0.95 at 0.03, 0.77 at 0.05,
# 0.63 at 0.07, 0.44 at 0.1.
TUNED_DT = 0.05
DT_SWEEP = (0.002, 0.02, 0.1)
BURN_IN = 300           # device steps before the timed window (set-up + warm-up): the chains' burn-in -- acceptance and the share of
                        # chains with anomalous dispersion are stationary by then (scripts/dt_sweep.py)
# configs[3]: where to time dual averaging.  The reference adapts the step size for `ndraws` trajectories and samples at the frozen
# dtbar afterwards (hmcda.py:329-345): the leg's window lies BEHIND the adaptation of (nearly) every chain.  On this noise-free
# 50-layer problem dual averaging drives the step sizes down to a few 1e-3 (a trajectory of fixed length lambda = L0 dt0 = 1 leaves
# the region where the root search succeeds whatever its step size, and every failure counts as alpha = 0), so a trajectory is
# 300-1000 device steps and DA_NDRAWS of them take ~10 000 steps: they run untimed, for at most DA_ADAPT_CAP device steps or
# DA_ADAPT_BUDGET_S seconds (the driver allows the bench 1 800 s), and the leg says what share of the chains was through.
DA_ADAPT_CAP = int(os.environ.get("RFS_DA_ADAPT_CAP", "60000"))     # device steps the configs[3] leg waits for 95 % of its chains to finish adapting (then it times anyway and says so)
DA_ADAPT_BUDGET_S = float(os.environ.get("RFS_DA_ADAPT_BUDGET_S", "420"))   # ... or this many seconds
DA_NDRAWS = int(os.environ.get("RFS_DA_NDRAWS", "20"))          # adapting trajectories per chain in the configs[3] leg (the reference's param.yaml: 200)
SUSTAIN_K = 100         # --steps below this: a second timed window of this many steps follows the contract's K (reported beside it)
SIDE_BURN, SIDE_K = 60, 100      # the side legs (they continue burned-in chains): untimed / timed device steps
DTYPE_TEXT = ("f64 (receiver-function row sweep beyond the Gaussian band: packed f32 where proven exact per chain, "
              "rf_f32_beyond_band)")
ROOT_MODE_TEXT = {
    "reference_roots": "inside a trajectory: warm start from the previous step's roots and kernels, branch test, then the "
                       "reference's own refinement inside its scan cell (swd_warm_start 1, swd_warm_exact 1): the reference's "
                       "float32 roots; start models and declined chains: the reference-semantics search",
    "converged_roots": "warm start + branch test only (swd_warm_exact 0): converged roots, within 1.1e-6 c of the reference's",
    "full_search": "the reference-semantics sequential search at every evaluation (swd_warm_start 0)",
}

CONFIGS = {
    1: dict(idx=1, n=30, nt=512, dt=0.1, sampler=None,
            name="configs[1]: 8192 chains x 30-layer Vs+thk, joint RF(512 samples)+SWD(40 Rc periods) per GPU"),
    4: dict(idx=4, n=30, nt=2048, dt=0.025, sampler=None, hmc_dt=0.025,      # (four times the RF samples: twice the curvature scale)
            name="configs[4]: frequency-domain RF with a 2048-point FFT, 8192 chains x 30 layers + 40 Rc periods per GPU"),
    3: dict(idx=3, n=50, nt=512, dt=0.1, sampler="da",
            name="configs[3]: main_DA.py dual averaging (HMCDualAveraging.sample_flow), 8192 chains x 50-layer model, "
                 "joint RF(512)+SWD(40 Rc) per GPU, per-chain dt and L"),
}
# SURVEY 8(d), Config 2's secondary run: configs[1] plus 40 Rayleigh GROUP periods (tRg = tRc): three root searches and three
# eigenfunction passes per group period (surfdisp.cpp:235-256, sregn96.f90:1747-1888) -- the costliest SWD shape
CONFIG1_RG = dict(CONFIGS[1], rg=True, idx="1rg",
                  name="configs[1] + 40 Rg periods (tRg = tRc = linspace(5, 44, 40)): 8192 chains x 30 layers, joint RF(512)+SWD(40 Rc + 40 Rg)")
# backwards-compatible module constants (scripts/ import them): the headline shape
N_LAYER, NT, DT = 30, 512, 0.1


def alg_bytes_per_eval(n, nt, nper=NPER):
    """plugin contract -- x in; misfit, grad, dsyn, flag out."""
    return 8 * 2 * n + 8 + 8 * 2 * n + 8 * (nt + nper) + 4          # 5388 at n = 30, nt = 512


def alg_flops_per_eval(n, nt):
    """Hand count of SURVEY.md section 8(d) for the minimal O(n) algorithm (each transcendental = 1 flop), split per
    kernel group; SWD scales with the layer count, RF with layers x frequencies.  n = 30, nt = 512: 2.8e7."""
    n2 = (1 << (nt - 1).bit_length()) // 2 + 1
    s, r = n / 30.0, (n / 30.0) * (n2 / 257.0)
    return {"swd_roots": 5.3e6 * s, "swd_eigen": 2.1e6 * s, "rf_pass_a": 0.6e7 * r, "rf_pass_b": 1.5e7 * r}


ALG_BYTES_PER_EVAL = alg_bytes_per_eval(N_LAYER, NT)
ALG_FLOPS_PER_EVAL = sum(alg_flops_per_eval(N_LAYER, NT).values())


# ------------------------------------------------------------------------------------------ the line the driver parses
LINE_LIMIT = 6000       # bytes: the driver keeps 8 KB of stdout; round 4's 21 KB line was cut and never parsed
DETAIL_FILE = "bench_detail.json"
_ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_step",
              "group_ms_per_step", "launches_per_step", "avg_launch_ms")
_CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "evals_per_s_per_core", "cpu_model", "shapes")
_CONFIG_KEYS = ("workload", "chains_per_gpu", "nlayer", "nt", "nper", "dt", "accept_ratio", "L_range", "sampler",
                "root_search_mode", "set_up_steps", "parallelism", "root_search_failures", "rf_f32_bins_share")


def _rnd(v, sig=7):
    """Floats to `sig` significant digits (the line is read by people and a parser, not fed back into arithmetic)."""
    if isinstance(v, float):
        return float(f"{v:.{sig}g}") if np.isfinite(v) else None
    if isinstance(v, dict):
        return {k: _rnd(x, sig) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_rnd(x, sig) for x in v]
    return v


def _clip(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 3] + "..."


def headline_line(res):
    """The ONE stdout line: the contract's keys, `roofline`, `cpu_baseline` and scalars only -- every nested leg, per-group
    table and prose note stays in bench_detail.json.  Always < LINE_LIMIT bytes."""
    out = {}
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data"):
        if k in res:
            out[k] = res[k]
    cfg = res.get("config") or {}
    out["config"] = {k: _clip(cfg[k], 160) for k in _CONFIG_KEYS if k in cfg}
    if res.get("roofline"):
        out["roofline"] = {k: res["roofline"][k] for k in _ROOF_KEYS if k in res["roofline"]}
    if res.get("roofline_fp64"):
        out["roofline_fp64"] = res["roofline_fp64"]
    if res.get("cpu_baseline"):
        cb = res["cpu_baseline"]
        out["cpu_baseline"] = {k: (_clip(cb[k], 200) if k != "shapes" else cb[k]) for k in _CPU_KEYS if k in cb}
    vi = (res.get("valu_issue") or {}).get("step")
    if vi:
        out["valu_issue_frac"] = vi.get("frac")
    rs = res.get("root_search") or {}
    for k, short in (("secular_evals_per_item_warm_start_and_branch_test", "evals_per_item_warm"),
                     ("secular_evals_per_item_reference_root_stage", "evals_per_item_exact"),
                     ("chains_handed_back_to_the_full_search_per_step", "handed_back_per_step")):
        if k in rs:
            out[short] = rs[k]
    for k, v in res.items():                       # every top-level scalar rides along (dt, accept_ratio, *_value, ...)
        if k not in out and not k.startswith("_") and isinstance(v, (int, float, bool, type(None))):
            out[k] = v
        elif k not in out and isinstance(v, str) and len(v) <= 40:
            out[k] = v
    out["detail"] = DETAIL_FILE
    out = _rnd(out)
    line = json.dumps(out, separators=(",", ":"))
    if len(line) >= LINE_LIMIT:                    # never again: drop the optional scalars before the contract's keys
        keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "roofline_fp64", "cpu_baseline", "detail")
        line = json.dumps({k: out[k] for k in keep if k in out}, separators=(",", ":"))
    assert len(line) < LINE_LIMIT, len(line)
    return line


def emit(res):
    """bench_detail.json (everything) beside bench.py -- and under gpurun_out/ where that exists -- then the headline line
    as the LAST line of stdout."""
    res = {k: v for k, v in res.items() if not k.startswith("_")}
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")) if not res.get("dry_run") else ():
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, DETAIL_FILE), "w") as f:
                    json.dump(res, f, indent=1)
        except OSError as e:
            print(f"bench.py: could not write {d}/{DETAIL_FILE}: {e}", file=sys.stderr)
    sys.stderr.flush()
    print(headline_line(res))
    sys.stdout.flush()


def true_model(n=N_LAYER):
    thk = np.full(n, 60.0 / n); thk[-1] = 0.0
    vs = np.linspace(2.8, 4.6, n)
    return np.hstack((vs, thk))


def bounds_of(x0):
    """Search range of main_base.py:64-77 / main_DA.py:64-77."""
    n = len(x0) // 2
    lo = np.r_[np.maximum(0.2 * x0[:n], 1.5), 0.8 * x0[n:]]
    hi = np.r_[np.minimum(1.8 * x0[:n], 5.0), 1.2 * x0[n:]]
    lo[-1], hi[-1] = 0.0, 2.0
    return np.stack([lo, hi], axis=1)


def make_models(nchain, seed, n=N_LAYER):
    """Sorted-prior initial models inside the main_base.py:64-77 bounds (pyhmc/hmc.py:74-93 rule)."""
    x0 = true_model(n)
    vs0, thk0 = x0[:n], x0[n:]
    lo = np.maximum(vs0 - 0.8 * vs0, 1.5); hi = np.minimum(vs0 + 0.8 * vs0, 5.0)
    rng = np.random.default_rng(seed)
    v = lo + (hi - lo) * rng.random((nchain, n))
    h = thk0 * (0.8 + 0.4 * rng.random((nchain, n)))
    h[:, -1] = 2.0 * rng.random(nchain)          # last thickness is a dummy in [0, 2]
    # hmc.py:85-93 sorts vs and carries the thicknesses along, then REJECTS a draw whose first 2n - 1 entries leave their
    # bounds (:95-99) -- with one thickness range for every layer the accepted draws are sorted velocities over
    # independent in-range thicknesses.  (Rounds 1-3 permuted the dummy last thickness, drawn in [0, 2], into the stack
    # as well: about half of the start models lay outside the bounds and were mirrored back by the first drift.)
    v = np.sort(v, axis=1)
    return np.hstack((v, h))


# ------------------------------------------------------------------------------------------ CPU baseline
def _cpu_worker(args):
    """One host core: the reference's own compiled sources (oracle/_ref: libsurf complete; RF propagator / partials
    core from RFModule.f90 + numpy irfft for the 15-line tail) driven by the oracle's numpy restatement of the
    plugins, one independent chain per process as the reference runs them (README.md:43-44).  Falls back to the C
    restatement where oracle/_ref is absent.  shape "joint": Joint_RF_SWD (configs[1] / [3] / [4]); "swd0": configs[0]'s
    SWD-only plugin (36 Rc + 36 Rg periods)."""
    wid, shape, n, nt, dt, dobs, xs, budget_s = args
    # (the reference's Fortran writes a model dump to unit 6 whenever its root search fails -- surfdisp96.f:320-336 -- and the
    # burned-in models of the bench do make it fail now and then: keep that out of the launcher's stdout, which carries the JSON line)
    try:
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    except OSError:
        pass
    try:                                   # numpy is already loaded (forked): pin its BLAS pool to this one core
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:
        pass
    try:
        os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[wid % len(os.sched_getaffinity(0))]})
    except Exception:
        pass
    from oracle import oracle as O
    if O.ref_available():
        kind, swd_lib, rf_lib = "reference", O.ref_libsurf(), O.RefRFCore()
    else:
        kind, swd_lib, rf_lib = "port", O.libsurf, O.librf
    if shape == "swd0":
        t0p = np.arange(5., 41.)
        model = O.SurfWD(tRc=t0p, tRg=t0p, lib=swd_lib)
        model.set_obsdata(dobs)
    else:
        t = np.linspace(5, 44, NPER)
        model = O.Joint_RF_SWD(1.0, 1.0, O.ReceiverFunc(RAY_P, nt, dt, GAUSS, TSHIFT, WATER, "P", "freq", lib=rf_lib),
                               O.SurfWD(tRc=t, lib=swd_lib))
        model.set_obsdata(dobs[:nt], dobs[nt:])
    model.misfit_and_grad(xs[0])                     # first call: library loading, page faults
    k = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        model.misfit_and_grad(xs[(wid + k) % len(xs)])
        k += 1
    return kind, k, time.perf_counter() - t0


def _cpu_quota():
    """CPU time this container may use, in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return q / per if q > 0 else None
    except Exception:
        return None


def _host_cpus():
    """(logical CPUs this process may use, physical cores among them, model name)."""
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    model, cores = "unknown", set()
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip(); break
    except Exception:
        pass
    try:
        out = subprocess.run(["lscpu", "-p=CPU,CORE,SOCKET"], capture_output=True, text=True, timeout=10).stdout
        for line in out.splitlines():
            if line and not line.startswith("#"):
                cpu, core, sock = (int(v) if v else 0 for v in line.split(",")[:3])
                if cpu in allowed:
                    cores.add((sock, core))
    except Exception:
        pass
    return allowed, (len(cores) if cores else len(allowed)), model


def _cpu_pool_rate(ncores, shape, n, nt, dt, dobs, xs, budget_s):
    import multiprocessing as mp
    ctx = mp.get_context("fork")
    sample = xs[:max(64, ncores)]
    with ctx.Pool(ncores) as pool:
        res = pool.map(_cpu_worker, [(w, shape, n, nt, dt, dobs, sample, budget_s) for w in range(ncores)])
    return res[0][0], float(sum(k / el for _, k, el in res)), int(sum(k for _, k, _ in res))


def cpu_baseline(cfg, xs, dobs, budget_s=15.0, others=None):
    """Reference CPU path on ALL host cores: one independent process per physical core (the reference's only
    parallelism: one chain per MPI rank), same models as the GPU run; runs in the launching process, which never
    initialised the GPU, after the GPU ranks have finished.  others: {"config3": dict(n, nt, dt, xs, dobs), ...} -- the
    other configurations' shapes, a few seconds each, reported under "shapes"."""
    from oracle import oracle as O
    O.build(ref=False)
    allowed, nphys, model = _host_cpus()
    quota = _cpu_quota()
    # one process per core the container can actually run: its physical cores, capped by the cgroup's CPU quota (more
    # processes than that only time-share the same cores)
    ncores = max(1, min(nphys, int(quota)) if quota else nphys)
    kind, total, nev = _cpu_pool_rate(ncores, "joint", cfg["n"], cfg["nt"], cfg["dt"], dobs, xs, budget_s)
    out = {"value": total, "unit": "evals/s", "cores": ncores, "physical_cores_visible": nphys,
           "logical_cpus": len(allowed), "cpu_quota_cores": quota, "cpu_model": model,
           "evals_per_s_per_core": total / ncores,
           "kind": kind,
           "kind_detail": ("reference (RF tail numpy): oracle/_ref = the reference's own sources, compiled in the BUILD "
                           "container by oracle/Makefile and shipped to this box as binaries -- libsurf complete, RF "
                           "propagator/partials core of RFModule.f90; the 15-line RF tail (water level, Gaussian, irfft, "
                           "e^{sigma t}) is numpy because FFTW3 is absent from the image" if kind == "reference"
                           else "port: the C restatement oracle/liboracle.so"),
           "sample": f"{nev} joint misfit+grad evaluations of the bench's own {cfg['n']}-layer "
                     f"models (nt = {cfg['nt']}, {NPER} Rc periods), {ncores} independent processes (one per core the "
                     f"container may use) for {budget_s:.0f} s each"}
    shapes = {}
    for tag, o in (others or {}).items():
        try:
            nc = 1 if o.get("shape") == "swd0" else ncores          # configs[0] is ONE chain on one core (param.yaml, mpi4py n = 1)
            _, rate, ne = _cpu_pool_rate(nc, o.get("shape", "joint"), o["n"], o["nt"], o["dt"], np.array(o["dobs"]), np.array(o["xs"]),
                                         o.get("budget_s", 5.0))
            shapes[tag] = {"value": rate, "cores": nc, "evals": ne}
        except Exception as e:                                        # (a side figure must not cost the line)
            shapes[tag] = {"value": None, "error": str(e)[:200]}
    if shapes:
        out["shapes"] = {k: (None if v["value"] is None else float(f"{v['value']:.5g}")) for k, v in shapes.items()}
        out["shapes_detail"] = shapes
    return out


# ------------------------------------------------------------------------------------------ launcher
def launch(args, argv):
    """Parent process: spawn N rank processes, pass rank 0's JSON line through (adding the CPU baseline at N = 1)."""
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   RFS_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # as torchrun does for its workers: a rank is one GPU's host thread, not an OpenMP team.  With one OpenMP thread
        # per visible CPU the idle spinners exhaust a cgroup CPU quota and the kernel freezes the rank for the rest of each
        # 100 ms period (see rfsurfhmc_amd.pyhmc._batched.host_threads)
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    # rank 0's stdout is drained by a thread; the children are polled: the first rank that dies takes the others with it
    # (a rank missing from the rendezvous would leave them waiting for ever) and its exit code is reported
    import threading
    buf = []
    th = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    th.start()
    deadline = time.time() + args.timeout
    rcs = [None] * n
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
        failed = [i for i, rc in enumerate(rcs) if rc not in (None, 0)]
        if failed or time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for i, p in enumerate(procs):
                try:
                    rcs[i] = p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill(); rcs[i] = p.wait()
            th.join(timeout=5)
            print("".join(buf), file=sys.stderr)
            raise SystemExit(f"bench.py: {'timeout' if not failed else 'rank(s) ' + str(failed) + ' failed'}; rank exit codes {rcs}")
        time.sleep(0.05)
    th.join()
    out0 = "".join(buf)
    line = None
    for ln in (out0 or "").splitlines():
        if ln.startswith("{"):
            line = ln
        else:
            print(ln, file=sys.stderr)
    if any(rcs) or line is None:
        print(out0 or "", file=sys.stderr)
        raise SystemExit(f"bench.py: rank exit codes {rcs}")
    res = json.loads(line)
    assert res["n_gpus"] == n, (res["n_gpus"], n)
    if n == 1 and not args.no_cpu_baseline and not args.dry_run:
        cfg = CONFIGS[args.config]
        side = res.pop("_cpu_inputs")
        res["cpu_baseline"] = cpu_baseline(cfg, np.array(side["xs"]), np.array(side["dobs"]), others=side.get("others"))
        for tag, v in (res["cpu_baseline"].get("shapes") or {}).items():
            if v and isinstance(res.get(f"{tag}_value"), (int, float)):
                res[f"{tag}_gpu_over_cpu"] = res[f"{tag}_value"] / v
        g = res["value"] / res["cpu_baseline"]["value"]
        res["gpu_over_cpu_node"] = g
        # the two components of that ratio.  The CPU figure is the reference as written (O(n^2)
        # propagator loop, 1 + 4 n inverse FFTs); the GPU path runs the O(n) adjoint algorithm.
        alg = 4.6e8 / ALG_FLOPS_PER_EVAL if cfg["n"] == N_LAYER and cfg["nt"] == NT else None
        res["cpu_baseline"]["speedup_split"] = {
            "algorithmic": alg, "hardware_and_implementation": g / alg if alg else None,
            "note": "algorithmic = flop count of the reference as written (4.6e8 per evaluation, SURVEY 8(d)) / the minimal O(n) "
                    "algorithm's (2.8e7); the rest is this box's GPU against its host cores.  A CPU build of the O(n) algorithm does "
                    "not ship (no CPU fallback in the product), so the split is by flop count, not by a second CPU measurement"}
    res.pop("_cpu_inputs", None)
    emit(res)


# ------------------------------------------------------------------------------------------ one rank
def dry_rank(args, rank, world):
    """No GPU: the launcher / process-group / gather plumbing only (CPU test of `--gpus N`, gloo)."""
    import torch
    import torch.distributed as dist
    from rfsurfhmc_amd.chains import gather_misfits, shard_range
    if world > 1:
        dist.init_process_group(backend="gloo")
        assert dist.get_world_size() == args.gpus
        dist.barrier()
    nchain = args.chains
    first, last = shard_range(nchain * world, rank, world)
    misfit = torch.arange(first, last, dtype=torch.float64)
    el = torch.tensor([0.001 * (rank + 1) * args.steps], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        gathered = gather_misfits(misfit)
    else:
        gathered = misfit
    if rank == 0:
        assert gathered.shape[0] == nchain * world and bool((gathered == torch.arange(nchain * world)).all())
        print(json.dumps({"metric": METRIC, "value": None, "unit": "evals/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": float(el) / args.steps * 1e3, "dry_run": True,
                          "gathered_chains": int(gathered.shape[0])}))
    if world > 1:
        dist.destroy_process_group()


CLOCK_HZ = 2.4e9        # MI355X peak engine clock (MI355X_MICROARCH.md)
N_SIMD = 256 * 4         # 256 CUs x 4 SIMDs; one f64 (or any VALU) wave-instruction occupies a SIMD for 4 clocks


def _counters(config):
    """Per-step PMC figures of a configuration (profiles/r*_counters.json, written by scripts/pmc_summary.py from
    separate rocprofv3 --pmc passes); None where no pass was committed."""
    for name in ("r06_counters.json", "r05_counters.json", "r04_counters.json", "r03_counters.json"):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", name)))
            if tj.get("chains") == 8192 and f"config{config}" in tj:
                return tj[f"config{config}"]
        except Exception:
            continue
    return None


def leg_report(cfg, config, nchain, K, el, evals, ms_step, launches_step, dom, dom_live_ms_launch, dom_live_ms_step=None):
    """The figures of one timed leg: rate, per-step kernel-group times, roofline of the group with the largest per-STEP
    sum, VALU issue per group from the committed counter passes."""
    n, nt = cfg["n"], cfg["nt"]
    ab = alg_bytes_per_eval(n, nt, NPER * (2 if cfg.get("rg") else 1))
    dom_ms = ms_step[dom] if dom_live_ms_step is None else dom_live_ms_step
    achieved = ab * nchain / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    cnt = _counters(config) if nchain == 8192 else None
    traffic = cnt[dom]["hbm_bytes"] if cnt and dom in cnt and cnt[dom]["hbm_bytes"] > 0 else None
    rep = {
        "ms_per_step": el / K * 1e3, "value": evals / el, "unit": "evals/s", "steps": K,
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_step": ab * nchain, "group_ms_per_step": dom_ms,
                     "launches_per_step": launches_step[dom], "avg_launch_ms": dom_live_ms_launch,
                     "note": "kernel group with the largest stand-alone time per step; achieved = algorithmic bytes of one step "
                             "/ its HIP-event time per step over the timed region; the path is FP64-VALU / transcendental "
                             "bound (SURVEY 8(d)): see valu_issue"},
        "kernel_ms_per_step": ms_step, "kernel_launches_per_step": launches_step,
    }
    if cnt:
        vi, tot = {}, 0.0
        for k, v in cnt.items():
            if k in ms_step and v.get("valu_insts", 0) > 0:
                full = v["valu_insts"] * 4.0 / N_SIMD / CLOCK_HZ * 1e3
                tot += full
                vi[k] = {"valu_wave_insts_per_step": v["valu_insts"], "ms_at_full_issue": full,
                         "measured_ms_per_step": ms_step[k], "frac": full / ms_step[k] if ms_step[k] > 0 else None}
        rep["valu_issue"] = {"per_group": vi, "step": {"ms_at_full_issue": tot, "measured_ms_per_step": el / K * 1e3,
                                                       "frac": tot / (el / K * 1e3)},
                             "note": "SQ_INSTS_VALU per leapfrog step (committed rocprofv3 --pmc passes, profiles/r0*_pmc_config*.csv) "
                                     "x 4 clocks / 1024 SIMDs / 2.4 GHz against the measured time; group times are HIP-event "
                                     "durations on two concurrent streams (they overlap, their sum exceeds the step)"}
    if cnt and any(v.get("fp64_flops", 0) > 0 for v in cnt.values()):
        # FP64 vector roofline from counters (the roofline that binds): SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 per
        # leapfrog step of the committed --pmc passes, FMA = 2 flops, the others 1, x 64 lanes per wave-instruction (an upper
        # estimate where lanes are masked off), against the step time measured HERE
        fl = {k: v["fp64_flops"] for k, v in cnt.items() if v.get("fp64_flops", 0) > 0}
        tot = float(sum(fl.values()))
        step_s = el / K
        rep["roofline_fp64"] = {"bound": "fp64_valu", "achieved_tflops": tot / step_s / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS,
                                "unit": "TFLOP/s", "frac": tot / step_s / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                                "fp64_flops_per_step": tot, "fp64_flops_per_eval": tot / nchain,
                                "per_group_flops_per_step": fl,
                                "note": "flops per step from the committed counter passes (same workload, 8192 chains) / this run's "
                                        "measured step time; counts every lane of a wave-instruction"}
    return rep


def make_joint(cfg, local_rank):
    from rfsurfhmc_amd.model.model_rf import ReceiverFunc
    from rfsurfhmc_amd.model.model_surf import SurfWD
    from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
    n, nt = cfg["n"], cfg["nt"]
    t = np.linspace(5, 44, NPER)
    joint = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(RAY_P, nt, cfg["dt"], GAUSS, TSHIFT, WATER, "P", "freq", device=local_rank),
                         SurfWD(tRc=t, tRg=t if cfg.get("rg") else None, device=local_rank))
    x_true = true_model(n)
    drf, dswd, flag = joint.forward(x_true)
    assert flag
    joint.set_obsdata(drf, dswd)
    for kv in filter(None, os.environ.get("RFS_OPTS", "").split(",")):      # (experiments: "name=value,...")
        k_, v_ = kv.split("="); joint._ensure(n).set_option(k_, int(v_))
    for opt, env in (("swd_exact_group", "RFS_EXACT_GROUP"), ("swd_exact_runup", "RFS_EXACT_RUNUP"), ("rf_mid_fused", "RFS_MID_FUSED"), ("swd_walk_dense", "RFS_WALK_DENSE")):      # (experiments)
        if os.environ.get(env):
            joint._ensure(n).set_option(opt, int(os.environ[env]))
    return joint, x_true, bounds_of(x_true)


class GroupTimer:
    """Per-kernel-group HIP-event times of the library (rfs_enable_timing / rfs_kernel_ms_sum)."""

    def __init__(self, ctx):
        from rfsurfhmc_amd._lib import K_NAMES
        self.ctx, self.names = ctx, K_NAMES

    def on(self, mask=None):
        self.ctx.check(self.ctx.L.rfs_synchronize(self.ctx.h))
        self.ctx.check(self.ctx.L.rfs_enable_timing(self.ctx.h, 1 if mask is None else 2 * mask))

    def off(self):
        self.ctx.check(self.ctx.L.rfs_enable_timing(self.ctx.h, 0))

    def read(self):
        ms = np.zeros(len(self.names)); cnt = np.zeros(len(self.names), dtype=np.int32)
        self.ctx.check(self.ctx.L.rfs_kernel_ms_sum(self.ctx.h, ms.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p)))
        return ms, cnt


def flow_leg(cfg, config, joint, x_true, bounds, nchain, rank, dev, K, nwarm, barrier, setup_steps=32):
    """configs[1] / configs[4]: K leapfrog steps of every chain on the flow entry (never-ending trajectories)."""
    import torch
    n = cfg["n"]
    ctx = joint._ensure(n)
    gt = GroupTimer(ctx)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    xs = make_models(nchain, seed=991206 + rank, n=n)      # chain c of rank r ~ reference rank r*nchain + c
    rng = np.random.default_rng(7 + rank)
    st = joint.flow_state(tt(xs), torch.full((nchain,), 0.002, dtype=torch.float64, device=dev), tt(bounds))
    st["p"].copy_(tt(0.5 * rng.standard_normal(xs.shape)))        # p ~ 0.5 N(0, I), hmc.py:146
    st["rem"].fill_(1 << 30); st["fresh"].fill_(1)
    joint.flow_step(st)                                            # start evaluation + half kick (untimed): full root search
    # set-up, not warm-up: the first steps hand the chains whose start thickness lay outside the bounds (mirrored back by
    # the first drift: a move no first-order model covers) to the full search once more; from then on every chain is continued
    for _ in range(setup_steps):
        joint.flow_step(st)
    gt.on()
    for _ in range(max(nwarm - 1, 1)):
        joint.flow_step(st)
    torch.cuda.synchronize()
    ms_w, cnt_w = gt.read()
    nw = max(nwarm - 1, 1)
    ms_step = {k: ms_w[i] / nw for i, k in enumerate(gt.names)}
    launches = {k: cnt_w[i] / nw for i, k in enumerate(gt.names)}
    dom_id = int(np.argmax([ms_step[k] if k != "flow_step" else 0.0 for k in gt.names]))
    gt.on(1 << dom_id)                                             # the dominant group only, measured live below
    d0 = ctx.stat("swd_warm_declined_chains"); i0 = ctx.stat("swd_warm_items"); e0 = ctx.stat("swd_warm_secular_evals")
    barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        joint.flow_step(st)
    ctx.check(ctx.L.rfs_synchronize(ctx.h))
    barrier()
    el = time.perf_counter() - t0
    ms, cnt = gt.read()
    gt.off()
    dom = gt.names[dom_id]
    ms_step[dom] = ms[dom_id] / K                                  # HIP events over the timed region
    launches[dom] = cnt[dom_id] / K
    dom_launch = ms[dom_id] / cnt[dom_id] if cnt[dom_id] else 0.0
    rep = leg_report(cfg, config, nchain, K, el, nchain * K, ms_step, launches, dom, dom_launch)
    items = ctx.stat("swd_warm_items") - i0
    rep["root_search"] = {
        "warm_started_items_per_step": items / K, "items_per_step": nchain * NPER,
        "secular_evals_per_item": (ctx.stat("swd_warm_secular_evals") - e0) / max(items, 1),
        "chains_handed_back_to_the_full_search_per_step": (ctx.stat("swd_warm_declined_chains") - d0) / K,
        "note": "inside a trajectory the roots of step s continue those of step s-1 (predictor from its Frechet kernels, "
                "bracket, false position, branch test: rfs_set_option swd_warm_start, default 1); the reference-semantics "
                "search runs for the start models and for every chain the continuation declines"}
    rep["root_search_failures"] = int((st["ok"] == 0).sum().item())
    rep["kernel_ms_note"] = (f"'{dom}' from HIP events over the timed region; the other groups from HIP events over the "
                             f"{nw} warm-up step(s) before it (all groups bracketed there)")
    return rep, st, xs, el


def set_root_mode(joint, n, mode):
    ctx = joint._ensure(n)
    joint.set_warm_start(0 if mode == "full_search" else 1)
    ctx.set_option("swd_warm_exact", 0 if mode == "converged_roots" else 1)


def sampler_leg(cfg, config, joint, x_true, bounds, nchain, rank, dev, K, burn, barrier, kind="hmc", dt=TUNED_DT,
                mode="reference_roots", xs=None, groups=True, K2=0, adapt_cap=0, nper_items=NPER):
    """A real sampler run on the continuous-flow schedule (HamitonianMC.sample_flow / HMCDualAveraging.sample_flow): `burn`
    device steps untimed, K timed (then, K2 > 0: a second timed window of K2 steps straight behind it -- the driver's K = 20 is a
    0.1 s window that starts from a drained device and ends waiting for the background searches).  kind "hmc": pyhmc/hmc.py:228-276 at step size dt, L ~ U{5..20} (param.yaml:38); "da":
    main_DA.py's dual averaging (dt0 0.1, L0 10, target 0.65).  groups: per-kernel-group times -- every group on its own in a
    few one-stream steps before the window (that picks the dominant group), the dominant one live over the window.
    adapt_cap > 0 (kind "da"): the window starts once 95 % of the chains have completed their `ndraws` adapting trajectories
    (hmcda.py:329-345: the step size is then frozen at dtbar) -- or after adapt_cap device steps, whichever comes first."""
    import torch
    from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
    from rfsurfhmc_amd.pyhmc.hmcda import HMCDualAveraging
    n = cfg["n"]
    set_root_mode(joint, n, mode)
    ctx = joint._ensure(n)
    gt = GroupTimer(ctx)
    if xs is None:
        if kind == "da":
            rs = np.random.default_rng(3 + rank)
            xs = np.clip(x_true[None, :] * (1 + 0.02 * rs.standard_normal((nchain, 2 * n))), bounds[:, 0], bounds[:, 1])
            xs[:, :n] = np.sort(xs[:, :n], axis=1)
        else:
            xs = make_models(nchain, seed=991206 + rank, n=n)      # chain c of rank r ~ reference rank r * nchain + c
    nsamp = (burn + K + K2) // 4 + 20                                   # more sample slots than trajectories can complete
    if kind == "da":
        # hmc block of the reference's param.yaml: dt 0.1, L0 10, target_ratio 0.65, seed 991206 (main_DA.py:79)
        # (the leg cannot afford param.yaml's 200 adapting trajectories of ~1 time unit each: DA_NDRAWS of them, the sample
        # count sized so that ndraws >= 0.1 nsamples holds, hmcda.py:57-60)
        nsamp = 10 * DA_NDRAWS if adapt_cap else nsamp
        smp = HMCDualAveraging(joint, bounds, 0.1, 10, 10, 0.65, 991206, nsamp, DA_NDRAWS if adapt_cap else 20, myrank=rank,
                               name="bench", outdir=None, nchains=nchain, verbose=False, store_syn=False)
        # every chain starts its first trajectory at the same step: the first ~3 trajectories (until dual averaging has
        # given the chains different step sizes) finish in bursts; time a window behind them
        burn = max(burn, 40)
    else:
        smp = HamitonianMC(joint, bounds, dt, [5, 20], 10, 991206, nsamp, 20, myrank=rank, name="bench", outdir=None,
                           nchains=nchain, verbose=False, store_syn=False)
    marks = {}
    STATS = ("swd_warm_declined_chains", "swd_warm_items", "swd_warm_secular_evals", "swd_exact_secular_evals",
             "swd_warm_walked_chains", "swd_exact_declined_chains", "flow_chain_steps", "swd_exact_evals_slowest_lane",
             "swd_exact_wavefronts")
    nser = 4 if (groups and burn >= 12) else 0

    class _StopLeg(Exception):
        pass

    sched = {"burn": None if (adapt_cap and kind == "da") else burn}

    def hook(s, st):
        burn = sched["burn"]
        if burn is None:                                           # dual averaging still adapting: look every 64 steps
            if s == 0:
                marks["adapt_t0"] = time.perf_counter()
            if s >= 40 and s % 64 == 0:
                share = float(np.mean(smp.live_counts[1] >= smp.ndraws))
                if share >= 0.95 or s >= adapt_cap or time.perf_counter() - marks["adapt_t0"] > DA_ADAPT_BUDGET_S:
                    sched["burn"] = s + nser + 3; marks["adapted_share"] = share
                    marks["adapt_s"] = time.perf_counter() - marks["adapt_t0"]
            return
        if nser and s == burn - nser - 2:
            ctx.set_option("swd_warm_serial", 1)                   # one stream: every kernel group alone on the chip
        if nser and s == burn - nser - 1:
            gt.on()
        if nser and s == burn - 1:
            marks["alone"] = gt.read()
            ctx.set_option("swd_warm_serial", 0)
            alone = np.array(marks["alone"][0], dtype=float)
            alone[[i for i, k in enumerate(gt.names) if k == "flow_step"]] = 0.0      # (the whole step's span is not a kernel group)
            marks["dom"] = int(np.argmax(alone))
            gt.on(1 << marks["dom"])                               # the dominant group only, measured live below
        if s == burn:
            marks["stat0"] = {k: ctx.stat(k) for k in STATS}
            marks["acc0"] = tuple(int(a.sum()) for a in smp.live_counts)
            barrier()
            if "dom" in marks:
                gt.read()                                          # (drop the step between the two phases)
            marks["t0"] = time.perf_counter(); marks["cpu0"] = time.process_time()
        if s == burn + K:
            marks["cpu1"] = time.process_time()
            ctx.check(ctx.L.rfs_synchronize(ctx.h))
            barrier()
            marks["t1"] = time.perf_counter()
            marks["stat1"] = {k: ctx.stat(k) for k in STATS}
            marks["acc1"] = tuple(int(a.sum()) for a in smp.live_counts)
            marks["dt"] = st["dt"].clone(); marks["rem"] = st["rem"].clone(); marks["x"] = st["x"].clone()
            marks["U"] = float(st["Ucur"].median().item()); marks["fail"] = int((st["ok"] == 0).sum().item())
            marks["misfit"] = st["Ucur"].clone()
            if "dom" in marks:
                marks["live"] = gt.read(); gt.off()             # the dominant group's HIP events over exactly the K timed steps
            marks["t1b"] = time.perf_counter()
            if not K2:
                raise _StopLeg
        if K2 and s == burn + K + K2:
            ctx.check(ctx.L.rfs_synchronize(ctx.h))
            barrier()
            marks["t2"] = time.perf_counter()
            marks["stat2"] = ctx.stat("flow_chain_steps")
            raise _StopLeg

    try:                                                           # (the hook ends the run once its last time stamp is taken)
        smp.sample_flow(x_init=xs, max_steps=(adapt_cap + 128 if sched["burn"] is None else burn) + K + K2 + 16, step_hook=hook)
    except _StopLeg:
        ctx.check(ctx.L.rfs_synchronize(ctx.h))
    burn = sched["burn"]
    el = marks["t1"] - marks["t0"]
    d = {k: marks["stat1"][k] - marks["stat0"][k] for k in STATS}
    evals = int(d["flow_chain_steps"])
    items = max(d["swd_warm_items"], 1)
    nacc, ntraj = marks["acc1"][0] - marks["acc0"][0], marks["acc1"][1] - marks["acc0"][1]
    if "dom" in marks:
        ms, cnt = marks["live"]
        ms_a, cnt_a = marks["alone"]
        ms_step = {k: ms_a[i] / nser for i, k in enumerate(gt.names)}
        launches = {k: cnt_a[i] / nser for i, k in enumerate(gt.names)}
        dom_id = marks["dom"]; dom = gt.names[dom_id]
        live_ms, live_n = ms[dom_id], cnt[dom_id]
        rep = leg_report(cfg, config, nchain, K, el, evals, ms_step, launches, dom, live_ms / live_n if live_n else 0.0,
                         dom_live_ms_step=live_ms / K)
        rep["kernel_ms_note"] = (f"kernel_ms_per_step: HIP-event durations of every group in {nser} one-stream steps before the "
                                 f"timed window (each kernel alone on the chip); roofline: '{dom}', the largest of them, by its "
                                 "HIP events over the timed region (two streams sharing the chip)")
    else:
        rep = {"ms_per_step": el / K * 1e3, "value": evals / el, "unit": "evals/s", "steps": K}
    dtv = marks["dt"].cpu().numpy()
    rep.update({
        "sampler": "HMCDualAveraging.sample_flow (dt0 0.1, L0 10, target_ratio 0.65)" if kind == "da" else
                   "HamitonianMC.sample_flow (L ~ U{5..20})",
        "dt": float(dt) if kind != "da" else None, "burn_in_steps": burn,
        "accept_ratio": nacc / max(ntraj, 1), "trajectories_completed": ntraj,
        "root_search_mode": mode, "chains_in_a_trajectory_per_step": evals / K,
        "misfit_median_at_the_end": marks["U"], "root_search_failures": marks["fail"],
        # CPU time of this rank's process (all its threads) per device step over the timed window: what 8 ranks on one node's
        # host cores have to find room for (a rank is one Python thread + the HIP runtime's helpers)
        "host_cpu_ms_per_step": (marks["cpu1"] - marks["cpu0"]) / K * 1e3,
        "root_search": {"warm_started_items_per_step": d["swd_warm_items"] / K, "items_per_step": nchain * nper_items,
                        "secular_evals_per_item_warm_start_and_branch_test": d["swd_warm_secular_evals"] / items,
                        "secular_evals_per_item_reference_root_stage": d["swd_exact_secular_evals"] / items,
                        "chains_walking_the_scan_grid_per_step": d["swd_warm_walked_chains"] / K,
                        "chains_handed_back_to_the_full_search_per_step": d["swd_warm_declined_chains"] / K,
                        "of_those_by_the_reference_root_stage_per_step": d["swd_exact_declined_chains"] / K,
                        # the reference-root stage's wavefronts execute what their slowest lane needs: evaluations needed / executed
                        "reference_root_stage_lane_efficiency": d["swd_exact_secular_evals"] / max(64.0 * d["swd_exact_evals_slowest_lane"], 1.0),
                        "reference_root_stage_evals_of_the_slowest_lane_per_wavefront": d["swd_exact_evals_slowest_lane"] / max(d["swd_exact_wavefronts"], 1)}})
    if K2:
        el2 = marks["t2"] - marks["t1b"]
        rep["sustained"] = {"steps": K2, "ms_per_step": el2 / K2 * 1e3,
                            "value": (marks["stat2"] - marks["stat1"]["flow_chain_steps"]) / el2, "unit": "evals/s"}
    if kind == "da":
        rep["L_cap"] = int(smp.L_cap); rep["trajectory_lengths_clamped_to_L_cap"] = int(getattr(smp, "n_L_clamped", 0))
        rep["adapting_trajectories_per_chain"] = int(smp.ndraws)
        if "adapted_share" in marks:
            rep["share_of_chains_past_adaptation_at_window_start"] = marks["adapted_share"]
            rep["adaptation_seconds_untimed"] = marks.get("adapt_s")
        # a length clamped to L_cap stands for int(lambda / dt) steps of the reference (hmcda.py:307, no cap there): the rate in
        # leapfrog steps is the same either way -- what the clamp changes is how many steps a trajectory is, not how fast one is
        Lw = np.maximum(1.0, np.floor(10 * 0.1 / dtv))
        rep["L_unclamped_mean"] = float(Lw.mean()); rep["L_clamped_mean"] = float(np.minimum(Lw, smp.L_cap).mean())
        rep["share_of_chains_above_L_cap"] = float((Lw > smp.L_cap).mean())
        Lv = np.maximum(1, (10 * 0.1 / dtv).astype(int))           # L = max(1, int(lambda / dt)), lambda = L0 * dt0 (hmcda.py:307)
        q = lambda a: [float(v) for v in np.quantile(a, [0.05, 0.5, 0.95])]
        rep["adapted_dt_quantiles_5_50_95"] = q(dtv); rep["adapted_dt_max"] = float(dtv.max())
        rep["L_quantiles_5_50_95"] = q(Lv)
    return rep, xs, el, evals, marks["x"].cpu().numpy(), marks.get("misfit")


def config0_leg(local_rank, dev, K=300, burn=40):
    """configs[0], the reference's param.yaml as it ships: ONE chain, the SWD-only plugin (model/model_surf.py), 10 layers,
    36 Rc + 36 Rg periods 5 .. 40 s (param.yaml:16-21), plain HMC at dt 0.1, L ~ U{5..20} (:36-40) -- the case the reference
    runs as `mpiexec -n 1`.  One chain is a latency measurement: evaluations per second of that single chain, ms per
    evaluation.  Returns (report, inputs of the CPU leg at this shape)."""
    import torch
    from rfsurfhmc_amd.model.model_surf import SurfWD
    from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
    thk = np.array([3., 3, 4, 5, 5, 6, 7, 8, 10, 0]); vs = np.linspace(2.9, 4.6, 10)
    t = np.arange(5., 41.)
    x0 = np.hstack((vs, thk))
    m = SurfWD(tRc=t, tRg=t, device=local_rank)
    d, flag = m.forward(x0)
    assert flag
    m.set_obsdata(d)
    bounds = bounds_of(x0)
    smp = HamitonianMC(m, bounds, 0.1, [5, 20], 10, 991206, 800, 200, myrank=0, name="bench0", outdir=None, nchains=1,
                       verbose=False, store_syn=False)
    ctx = m._ensure(10)
    marks = {}

    class _Stop(Exception):
        pass

    def hook(s, st):
        if s == burn:
            torch.cuda.synchronize(); marks["e0"] = ctx.stat("flow_chain_steps"); marks["a0"] = tuple(int(a.sum()) for a in smp.live_counts)
            marks["t0"] = time.perf_counter()
        if s == burn + K:
            ctx.check(ctx.L.rfs_synchronize(ctx.h)); torch.cuda.synchronize()
            marks["t1"] = time.perf_counter(); marks["e1"] = ctx.stat("flow_chain_steps")
            marks["a1"] = tuple(int(a.sum()) for a in smp.live_counts)
            raise _Stop

    try:
        smp.sample_flow(max_steps=burn + K + 8, step_hook=hook)
    except _Stop:
        pass
    el, ev = marks["t1"] - marks["t0"], marks["e1"] - marks["e0"]
    ntraj = marks["a1"][1] - marks["a0"][1]
    rep = {"workload": "configs[0]: param.yaml default -- 1 chain, 10-layer Vs model, SWD-only plugin (36 Rc + 36 Rg periods), "
                       "HamitonianMC dt 0.1, L ~ U{5..20}", "value": ev / el, "unit": "evals/s", "ms_per_eval": el / max(ev, 1) * 1e3,
           "device_steps": K, "evaluations": int(ev), "accept_ratio": (marks["a1"][0] - marks["a0"][0]) / max(ntraj, 1),
           "note": "one chain = one wavefront's worth of work per kernel: a latency figure, set by the length of the dependent "
                   "chain of one evaluation (three root searches + eigenfunction passes per group-velocity period)"}
    rng = np.random.default_rng(5)
    xs = np.clip(x0[None, :] * (1 + 0.02 * rng.standard_normal((16, 20))), bounds[:, 0], bounds[:, 1])
    xs[:, :10] = np.sort(xs[:, :10], axis=1)
    m._ctx.close(); m._ctx = None
    return rep, {"shape": "swd0", "n": 10, "nt": 0, "dt": 0.0, "xs": xs.tolist(), "dobs": np.asarray(d).tolist(), "budget_s": 4.0}


STEP_TEXT = {
    None: "one device step of HamitonianMC.sample_flow: one leapfrog step of every chain via rfs_flow_step2 (drift + mirror, "
          "misfit+gradient, kick; accept / reject and the next trajectory's start on the device from draws the host made ahead "
          "from every chain's MT19937 stream); value counts only chains inside a trajectory",
    "da": "one device step of HMCDualAveraging.sample_flow (accept / reject and restart on the device from draws made "
          "ahead, dual averaging on the host beside the steps); value counts only chains inside a trajectory",
}


def run_rank(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {world}")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.dry_run:
        return dry_rank(args, rank, world)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP extension has no CPU fallback")
    # RFS_BENCH_SHARED_GPU=1: functional check of the N > 1 code path on a box with ONE GPU -- every rank on device 0,
    # collectives over gloo (RCCL refuses two ranks on one device).  The number it prints is meaningless and says so.
    shared = os.environ.get("RFS_BENCH_SHARED_GPU") == "1" and world > 1
    if shared:
        local_rank = 0
    elif torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank}, the box has {torch.cuda.device_count()}")
    torch.cuda.set_device(local_rank)                # before the process group: RCCL binds to the current device
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if shared else dev    # where the collectives' tensors live
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="gloo" if shared else "nccl")      # "nccl" is RCCL on ROCm
        assert dist.get_world_size() == args.gpus

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    cfg = CONFIGS[args.config]
    n, nt = cfg["n"], cfg["nt"]
    nchain = args.chains
    K, W = args.steps, args.warmup
    burn = max(W, BURN_IN)                       # set-up steps (burn-in) + the W warm-up steps
    dt = args.dt if args.dt is not None else cfg.get("hmc_dt", TUNED_DT)
    mode = {None: "reference_roots", 1: "reference_roots", 0: "full_search"}[args.warm_start]
    if args.converged_roots:
        mode = "converged_roots"
    kind = "da" if cfg["sampler"] == "da" else "hmc"
    # --seed-rank R (single-rank jobs): run the chains rank R of a multi-rank job would run (tests)
    srank = rank if args.seed_rank is None else args.seed_rank
    joint, x_true, bounds = make_joint(cfg, local_rank)
    ctx = joint._ensure(n)
    extra = {}
    rep, xs, el, evals_rank, x_end, misfit = sampler_leg(cfg, args.config, joint, x_true, bounds, nchain, srank, dev, K, burn,
                                                          barrier, kind=kind, dt=dt, mode=mode,
                                                          K2=(SUSTAIN_K if K < SUSTAIN_K else 0),
                                                          adapt_cap=DA_ADAPT_CAP if kind == "da" else 0)
    if misfit is None:
        misfit = torch.zeros(nchain, dtype=torch.float64, device=dev)
    f32_chains = ctx.stat("rf_f32_chains")       # chain evaluations whose frequencies beyond the band were swept in float32
    f32_share = (1.0 - ctx.stat("rf_band_bins") / max(ctx.stat("rf_bins"), 1)) if f32_chains > 0 else 0.0
    side_legs = rank == 0 and world == 1 and kind == "hmc" and not args.headline_only
    if side_legs:
        # ---- the same chains, continued from where the headline left them (burned in), under other settings: short legs
        sb, sk = SIDE_BURN, SIDE_K

        def short(tag, **kw):
            r, _, _, _, _, _ = sampler_leg(cfg, args.config, joint, x_true, bounds, nchain, srank, dev, sk, sb, barrier,
                                           kind="hmc", xs=x_end, groups=False, **kw)
            r["note"] = tag
            return r

        extra["dt_sweep"] = [short("step-size sweep: the headline's chains, burned in, continued at this step size "
                                   "(param.yaml:40 uses 0.1)", dt=v, mode=mode) for v in DT_SWEEP]
        if mode == "reference_roots":
            extra["converged_roots"] = short(ROOT_MODE_TEXT["converged_roots"] + " -- misfits to ~1.4e-5, gradients to ~1e-5 "
                                             "except on ill-conditioned chains: outside the 1e-5 contract, kept as an option",
                                             dt=dt, mode="converged_roots")
            extra["full_search_every_step"] = short(ROOT_MODE_TEXT["full_search"], dt=dt, mode="full_search")
            # ONE run-up period per group, origins accepted to 5e-7 c: a sixth less work in the reference-root stage; the default's
            # parity figures on this regime (phase velocities), looser where group velocities difference the roots (rfsurf.h)
            ctx.set_option("swd_exact_runup", 1); ctx.set_option("swd_exact_origin_tol_e9", 500)
            extra["one_runup_period"] = short("swd_exact_runup 1, swd_exact_origin_tol_e9 500: an option (looser with group velocities)",
                                              dt=dt, mode=mode)
            ctx.set_option("swd_exact_runup", 2); ctx.set_option("swd_exact_origin_tol_e9", 100)
            # ... and the headline's own setting by the side legs' protocol, straight after it: a side leg continues chains that
            # have been through ~1 000 more steps than the headline's window and times a shorter window, so its rate compares with
            # THIS figure, not with `value`
            extra["headline_again"] = short("the headline's setting, measured like a side leg (for the side legs' ratios)",
                                            dt=dt, mode=mode)
            # ... and with EVERY frequency of the receiver-function row sweep in f64 (the reference's own arithmetic type
            # throughout): the headline's pass A takes the bins beyond the Gaussian band in packed float32 where a per-chain
            # proof allows it -- same trace, misfit and gradient to 3e-13 (tests/test_gpu_parity.py), fewer instructions
            ctx.set_option("rf_f32_beyond_band", 0)
            extra["all_f64"] = short("rf_f32_beyond_band 0: every frequency in f64; compares with headline_again", dt=dt, mode=mode)
            ctx.set_option("rf_f32_beyond_band", 1)
        set_root_mode(joint, n, mode)
        # ---- rounds 1-3's headline definition, for continuity: never-ending trajectories (no accept / reject) of the random
        # start models at dt = 0.002, the cheapest point of the step-size curve
        r0, st0, _, _ = flow_leg(cfg, args.config, joint, x_true, bounds, nchain, srank, dev, 40, 3, barrier)
        extra["never_ending_dt0002"] = {k: r0[k] for k in ("ms_per_step", "value", "unit", "steps", "kernel_ms_per_step")}
        extra["never_ending_dt0002"]["note"] = ("rfs_flow_step on never-ending trajectories from the random start models, dt = 0.002, "
                                                "33 set-up steps: the definition of rounds 1-3's headline (2.57 M evals/s in round 3 "
                                                "with converged roots; this figure: " + mode + ")")
        tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        xfix = tt(x_end)
        for _ in range(3):
            joint.misfit_and_grad_device(xfix)
        ctx.check(ctx.L.rfs_synchronize(ctx.h)); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(20):
            joint.misfit_and_grad_device(xfix)
        ctx.check(ctx.L.rfs_synchronize(ctx.h)); torch.cuda.synchronize()
        extra["eval_only_ms_per_call"] = (time.perf_counter() - t1) / 20 * 1e3      # plugin entry: full search every call

    total_evals = evals_rank
    sus = rep.get("sustained")
    sus_el = sus["ms_per_step"] * sus["steps"] * 1e-3 if sus else 0.0
    sus_evals = sus["value"] * sus_el if sus else 0.0
    if dist is not None:
        red = torch.tensor([el, sus_el, float(evals_rank), sus_evals], dtype=torch.float64, device=cdev)
        tmax = red[:2].clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = red[2:].clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        el, total_evals = float(tmax[0].item()), float(tsum[0].item())
        sus_el, sus_evals = float(tmax[1].item()), float(tsum[1].item())
        # the path's only collective: gather the per-chain misfits on rank 0 (after the timed region)
        from rfsurfhmc_amd.chains import gather_misfits
        gathered = gather_misfits(misfit.to(cdev))
        assert rank != 0 or gathered.shape[0] == nchain * world
        if rank == 0 and os.environ.get("RFS_BENCH_DUMP"):
            np.save(os.environ["RFS_BENCH_DUMP"], gathered.cpu().numpy())
    elif os.environ.get("RFS_BENCH_DUMP"):
        np.save(os.environ["RFS_BENCH_DUMP"], misfit.cpu().numpy())
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- the other single-GPU configurations of BASELINE.json, as short legs of the default run
    cpu_others = {}
    if world == 1 and args.config == 1 and nchain == 8192 and not args.headline_only and not args.no_other_configs:
        joint._ctx.close(); joint._ctx = None; joint._cfg = None
        torch.cuda.empty_cache()
        for ci in ("1_rg", 4, 3):
            c2 = CONFIG1_RG if ci == "1_rg" else CONFIGS[ci]
            j2 = None
            try:                                     # (a side leg must not cost the line)
                j2, xt2, b2 = make_joint(c2, local_rank)
                k2 = "da" if c2["sampler"] == "da" else "hmc"
                r2, _, _, _, xe2, _ = sampler_leg(c2, ci, j2, xt2, b2, 8192, rank, dev, SIDE_K, BURN_IN, barrier, kind=k2,
                                                  dt=c2.get("hmc_dt", TUNED_DT), mode=mode, adapt_cap=DA_ADAPT_CAP if k2 == "da" else 0,
                                                  nper_items=NPER * (4 if c2.get("rg") else 1))
                r2["workload"] = c2["name"]; r2["step"] = STEP_TEXT[c2["sampler"]]
                extra[f"config{ci}"] = r2
                if ci != "1_rg":
                    cpu_others[f"config{ci}"] = {"shape": "joint", "n": c2["n"], "nt": c2["nt"], "dt": c2["dt"], "xs": xe2[:64].tolist(),
                                                 "dobs": j2.dobs.tolist(), "budget_s": 5.0}
            except Exception as e:
                extra[f"config{ci}"] = {"value": None, "error": repr(e)[:300]}
                print(f"bench.py: the configs[{ci}] leg failed: {e!r}", file=sys.stderr)
            if j2 is not None and j2._ctx is not None:
                j2._ctx.close(); j2._ctx = None
            torch.cuda.empty_cache()
        try:
            extra["config0"], cpu_others["config0"] = config0_leg(local_rank, dev)
        except Exception as e:                       # (a side leg must not cost the line)
            extra["config0"] = {"value": None, "error": repr(e)[:300]}

    value = total_evals / el
    res = {
        "metric": METRIC,
        "value": value, "unit": "evals/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": el / K * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64 (+f32 beyond band)" if f32_chains > 0 else "f64", "dtype_note": DTYPE_TEXT,
        "data": "synthetic",
        "config": {"workload": cfg["name"] if not (args.config == 1 and world * nchain == 65536) else
                   "configs[2]: 65536 chains x 30-layer joint RF+SWD, 8xMI355X independent-chain shard, RCCL gather "
                   "(= configs[1]'s 8192 chains on each GPU)", "chains_per_gpu": nchain, "nlayer": n, "nt": nt, "nper": NPER,
                   "set_up_steps": burn - W, "set_up_note": f"untimed set-up: the chains are burned in for {burn - W} device steps "
                   f"from their random start models before the {W} warm-up steps (burn-in = {burn} steps in all)",
                   "sampler": rep["sampler"], "dt": rep["dt"], "dt_note": (None if kind == "da" else
                   f"step size {dt}: tuned for an acceptance ratio within 0.65-0.9 (SURVEY 8(d)); the reference's default 0.1 "
                   "(param.yaml:40) gives 0.44 here -- see dt_sweep" if args.dt is None else "--dt"),
                   "accept_ratio": rep["accept_ratio"], "L_range": [5, 20] if kind == "hmc" else None,
                   "root_search_mode": mode, "root_search_mode_text": ROOT_MODE_TEXT[mode],
                   "step": STEP_TEXT[cfg["sampler"]],
                   "parallelism": f"independent chains x{world}" + (" -- FUNCTIONAL CHECK: all ranks share GPU 0, gloo "
                                                                    "collectives; not a measurement" if shared else ""),
                   "root_search_failures": rep.get("root_search_failures", 0),
                   "rf_f32_bins_share": f32_share},
        # scalars the driver's parser keeps
        "dt": rep["dt"], "accept_ratio": rep["accept_ratio"], "root_search_mode": mode,
        "roofline": rep.get("roofline"),
    }
    if rep.get("roofline_fp64"):
        res["roofline_fp64"] = {k: rep["roofline_fp64"][k] for k in ("bound", "achieved_tflops", "peak", "unit", "frac", "fp64_flops_per_eval")}
    for k in ("kernel_ms_per_step", "kernel_launches_per_step", "valu_issue", "root_search", "kernel_ms_note", "host_cpu_ms_per_step",
              "chains_in_a_trajectory_per_step", "misfit_median_at_the_end", "trajectories_completed",
              "adapted_dt_quantiles_5_50_95", "adapted_dt_max", "L_quantiles_5_50_95", "L_cap", "trajectory_lengths_clamped_to_L_cap",
              "adapting_trajectories_per_chain", "share_of_chains_past_adaptation_at_window_start", "burn_in_steps"):
        if k in rep:
            res[k] = rep[k]
    if sus:
        # (--steps below SUSTAIN_K: the contract's K-step window is `value`; the longer window right behind it rides along)
        res["sustained_steps"] = sus["steps"]; res["sustained_ms_per_step"] = sus_el / sus["steps"] * 1e3
        res["sustained_value"] = sus_evals / sus_el
    for k, v in extra.items():
        res[k] = v
    # the side legs' rates as top-level scalars as well
    for k in ("converged_roots", "full_search_every_step", "one_runup_period", "headline_again", "all_f64", "never_ending_dt0002", "config3",
              "config4", "config0", "config1_rg"):
        if k in extra:
            res[f"{k}_value"] = extra[k]["value"]
    if "config0" in extra and extra["config0"].get("ms_per_eval"):
        res["config0_ms_per_eval"] = extra["config0"]["ms_per_eval"]
    if "config3" in extra:
        res["config3_accept_ratio"] = extra["config3"].get("accept_ratio")
        res["config3_adapted_share"] = extra["config3"].get("share_of_chains_past_adaptation_at_window_start")
    if "config1_rg" in extra:
        res["config1_rg_accept_ratio"] = extra["config1_rg"].get("accept_ratio")
    for r in extra.get("dt_sweep", []):
        tag = str(r["dt"]).replace(".", "p")
        res[f"dt_{tag}_value"] = r["value"]; res[f"dt_{tag}_accept_ratio"] = r["accept_ratio"]
    if world == 1 and not args.no_cpu_baseline and os.environ.get("RFS_BENCH_CHILD"):
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        res["_cpu_inputs"] = {"xs": x_end[:max(64, ncpu)].tolist(), "dobs": joint.dobs.tolist(), "others": cpu_others}
    if os.environ.get("RFS_BENCH_CHILD"):
        print(json.dumps(res))                       # to the launching process (a pipe), which adds the CPU baseline and emits
        sys.stdout.flush()
    else:
        emit(res)                                    # a rank of torch.distributed.run: the line itself
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed device steps of the sampler")
    ap.add_argument("--warmup", type=int, default=300, help=f"untimed warm-up steps before them; the chains are burned in for max(0, {BURN_IN} - W) set-up steps before that")
    ap.add_argument("--dt", type=float, default=None, help=f"HMC step size (default {TUNED_DT}: acceptance within 0.65-0.9)")
    ap.add_argument("--converged-roots", action="store_true", help="rfs_set_option swd_warm_exact 0 for the headline")
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS))
    ap.add_argument("--chains", type=int, default=8192, help="chains per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-run", action="store_true", help="launcher / process-group plumbing only (no GPU; gloo)")
    ap.add_argument("--seed-rank", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--timeout", type=float, default=1500.0, help="seconds before the launcher gives up on its ranks")
    ap.add_argument("--headline-only", action="store_true", help="only the timed headline leg (profiling runs)")
    ap.add_argument("--no-sampler-leg", action="store_true", help=argparse.SUPPRESS)      # (accepted for old command lines)
    ap.add_argument("--sustain", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--no-other-configs", action="store_true", help="skip the configs[3] / configs[4] legs of the default run")
    ap.add_argument("--warm-start", type=int, default=None, choices=[0, 1], help="rfs_set_option swd_warm_start (default: library's, 1)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" in os.environ:                   # a rank: ours (RFS_BENCH_CHILD) or torch.distributed.run's
        return run_rank(args)
    launch(args, sys.argv[1:])                       # the launching process never initialises the GPU


if __name__ == "__main__":
    main()
