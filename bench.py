#!/usr/bin/env python3
"""bench.py -- leapfrog-step (= misfit+gradient evaluation) throughput of the HIP hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--chains C]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): 8192 chains x 30-layer Vs+thk models per GPU, joint
RF (P, nt=512, dt=0.1, Gaussian 1.5, shift 5 s, water 1e-3, freq method) + 40 Rayleigh phase
periods linspace(5,44,40); synthetic sorted-prior models, dobs = forward(true model).
A "step" = one evaluation of misfit+gradient for every chain of the rank (what one leapfrog step
of every chain costs); inputs are resident in HBM when the timed region starts.
Independent chains shard across ranks (weak scaling, no data-path collective); the only
collective is the RCCL gather of the per-chain misfits after the timed region.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_LAYER, NT, DT, NPER = 30, 512, 0.1, 40
RAY_P, GAUSS, TSHIFT, WATER = 0.045, 1.5, 5.0, 0.001
# algorithmic bytes / flops per evaluation at this shape (plugin contract:
# x in; misfit, grad, dsyn, flag out)
ALG_BYTES_PER_EVAL = 8 * 2 * N_LAYER + 8 + 8 * 2 * N_LAYER + 8 * (NT + NPER) + 4     # 5388
ALG_FLOPS_PER_EVAL = 2.8e7
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s HBM3E
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X vector FP64


def true_model():
    thk = np.full(N_LAYER, 2.0); thk[-1] = 0.0
    vs = np.linspace(2.8, 4.6, N_LAYER)
    return np.hstack((vs, thk))


def make_models(nchain, seed):
    """Sorted-prior initial models inside the main_base.py:64-77 bounds (pyhmc/hmc.py:74-93 rule)."""
    x0 = true_model()
    vs0, thk0 = x0[:N_LAYER], x0[N_LAYER:]
    lo = np.maximum(vs0 - 0.8 * vs0, 1.5); hi = np.minimum(vs0 + 0.8 * vs0, 5.0)
    rng = np.random.default_rng(seed)
    v = lo + (hi - lo) * rng.random((nchain, N_LAYER))
    h = thk0 * (0.8 + 0.4 * rng.random((nchain, N_LAYER)))
    h[:, -1] = 2.0 * rng.random(nchain)          # last thickness is a dummy in [0, 2]
    idx = np.argsort(v, axis=1)
    v = np.take_along_axis(v, idx, axis=1)
    h[:, :-1] = np.take_along_axis(h, idx, axis=1)[:, :-1]
    return np.hstack((v, h))


def cpu_baseline(xs, dobs, budget_s=12.0):
    """Reference CPU path timed on one host core: the reference's own compiled sources (oracle/_ref:
    libsurf complete; RF propagator/partials core + numpy irfft tail) driven by the oracle's numpy
    restatement of the plugins; falls back to the C restatement when oracle/_ref is absent."""
    from oracle import oracle as O
    O.build(ref=False)
    t = np.linspace(5, 44, NPER)
    if O.ref_available():
        kind, swd_lib, rf_lib = "reference", O.ref_libsurf(), O.RefRFCore()
    else:
        kind, swd_lib, rf_lib = "port", O.libsurf, O.librf
    joint = O.Joint_RF_SWD(1.0, 1.0, O.ReceiverFunc(RAY_P, NT, DT, GAUSS, TSHIFT, WATER, "P", "freq", lib=rf_lib),
                           O.SurfWD(tRc=t, lib=swd_lib))
    joint.set_obsdata(dobs[:NT], dobs[NT:])
    n = 0
    t0 = time.perf_counter()
    while n < len(xs) and time.perf_counter() - t0 < budget_s:
        joint.misfit_and_grad(xs[n])
        n += 1
    el = time.perf_counter() - t0
    return {"value": n / el, "unit": "evals/s", "cores": 1, "kind": kind,
            "sample": f"{n} joint misfit+grad evaluations of the bench's own 30-layer models on 1 host core in {el:.1f} s"
                      + ("; reference = oracle/_ref (libsurf complete, RF core + numpy irfft tail)" if kind == "reference" else "")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--chains", type=int, default=8192, help="chains per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP extension has no CPU fallback")
    torch.cuda.set_device(local_rank)                # before the process group: RCCL binds to the current device
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl")      # RCCL on ROCm

    from rfsurfhmc_amd._lib import K_NAMES
    from rfsurfhmc_amd.model.model_rf import ReceiverFunc
    from rfsurfhmc_amd.model.model_surf import SurfWD
    from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD

    nchain = args.chains
    t = np.linspace(5, 44, NPER)
    joint = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(RAY_P, NT, DT, GAUSS, TSHIFT, WATER, "P", "freq", device=local_rank),
                         SurfWD(tRc=t, device=local_rank))
    drf, dswd, flag = joint.forward(true_model())
    assert flag
    joint.set_obsdata(drf, dswd)
    xs = make_models(nchain, seed=991206 + rank)          # chain c of rank r ~ reference rank r*nchain + c
    x = torch.from_numpy(xs).to(dev)
    ctx = joint._ensure(N_LAYER)

    # Warm-up (untimed steps).  From the second one on every kernel group is bracketed by HIP events: that gives the
    # per-group table of the bench line and names the dominant group; inside the timed region only THAT group keeps
    # its event pairs (each pair costs a few microseconds of the step: 0.1 ms with all seven groups).
    nwarm = max(args.warmup, 2)
    out = joint.misfit_and_grad_device(x)
    ctx.check(ctx.L.rfs_synchronize(ctx.h))
    ctx.check(ctx.L.rfs_enable_timing(ctx.h, 1))
    for _ in range(nwarm - 1):
        out = joint.misfit_and_grad_device(x)
    torch.cuda.synchronize()
    ms_w = np.zeros(len(K_NAMES)); cnt_w = np.zeros(len(K_NAMES), dtype=np.int32)
    ctx.check(ctx.L.rfs_kernel_ms_sum(ctx.h, ms_w.ctypes.data_as(ctypes.c_void_p), cnt_w.ctypes.data_as(ctypes.c_void_p)))
    warm_ms = {k: (ms_w[i] / cnt_w[i] if cnt_w[i] else 0.0) for i, k in enumerate(K_NAMES)}
    dom_id = int(np.argmax([warm_ms[k] for k in K_NAMES]))
    ctx.check(ctx.L.rfs_enable_timing(ctx.h, 2 * (1 << dom_id)))     # the dominant group only, measured live below
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = joint.misfit_and_grad_device(x)
    ctx.check(ctx.L.rfs_synchronize(ctx.h))
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    ms = np.zeros(len(K_NAMES)); cnt = np.zeros(len(K_NAMES), dtype=np.int32)
    ctx.check(ctx.L.rfs_kernel_ms_sum(ctx.h, ms.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p)))
    ctx.check(ctx.L.rfs_enable_timing(ctx.h, 0))
    misfit, grad, dsyn, fl = out
    nfail = int((fl == 0).sum().item())
    if dist is not None:
        tmax = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        el = float(tmax.item())
        # the path's only collective: gather the per-chain misfits on rank 0 (after the timed region)
        from rfsurfhmc_amd.chains import gather_misfits
        gathered = gather_misfits(misfit)
        assert rank != 0 or gathered.shape[0] == nchain * world
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    evals = nchain * world * args.steps
    value = evals / el
    dom = K_NAMES[dom_id]
    dom_ms = ms[dom_id] / cnt[dom_id] if cnt[dom_id] else 0.0       # HIP events over the timed region
    per_launch_ms = dict(warm_ms)                                   # the other groups: from the warm-up steps
    per_launch_ms[dom] = dom_ms
    achieved = ALG_BYTES_PER_EVAL * nchain / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    # HBM bytes per launch of that kernel group from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); only
    # valid for the configuration it was collected on
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        if tj.get("chains") == nchain and dom in tj:
            traffic = tj[dom]
    except Exception:
        traffic = None
    res = {
        "metric": "leapfrog steps/sec (= forward+grad evals/sec) per GPU and whole node, 30-layer model",
        "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "configs[1]: 8192 chains x 30-layer Vs+thk, joint RF(512 samples)+SWD(40 Rc periods) per GPU",
                   "chains_per_gpu": nchain, "nlayer": N_LAYER, "nt": NT, "nper": NPER,
                   "parallelism": f"independent chains x{world}", "root_search_failures": nfail},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": ALG_BYTES_PER_EVAL * nchain,
                     "avg_launch_ms": dom_ms,
                     "note": "path is FP64-VALU/transcendental bound (SURVEY 8(d)); see fp64_vector"},
        "fp64_vector": {"achieved_tflops": ALG_FLOPS_PER_EVAL * value / world / 1e12, "peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                        "frac": ALG_FLOPS_PER_EVAL * value / world / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                        "alg_flops_per_eval": ALG_FLOPS_PER_EVAL},
        "kernel_ms_per_launch": per_launch_ms,
        "kernel_ms_note": f"'{dom}' from HIP events over the timed region; the other groups from HIP events over the "
                          f"{nwarm - 1} warm-up step(s) before it (all groups bracketed there)",
    }
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(xs, joint.dobs)
    print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
