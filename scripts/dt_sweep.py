"""Acceptance ratio and leapfrog rate of HamitonianMC.sample_flow on the bench's chains as a function of the step size
(param.yaml:40 uses dt = 0.1; SURVEY 8(d): "tune so accept ~ 0.65-0.9").   python3 scripts/dt_sweep.py [steps=300] dt..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
W0 = int(os.environ.get("DT_SWEEP_BURN", "25"))      # device steps before the timed window
dts = [float(v) for v in sys.argv[2:]] or [0.002, 0.005, 0.01, 0.02, 0.05, 0.1]
cfg = bench.CONFIGS[1]
nchain = 8192
for dt in dts:
    joint, x_true, bounds = bench.make_joint(cfg, 0)
    ctx = joint._ensure(cfg["n"])
    xs = bench.make_models(nchain, 991206, cfg["n"])
    smp = HamitonianMC(joint, bounds, dt, [5, 20], 10, 991206, 2000, 20, myrank=0, name="bench", outdir=None,
                       nchains=nchain, verbose=False, store_syn=False)
    mk = {}
    smp_box = [smp]
    def hook(s, st):
        if s == W0:
            mk["f0"] = ctx.stat("flow_chain_steps"); mk["d0"] = ctx.stat("swd_warm_declined_chains")
            mk["i0"], mk["e0"], mk["x0"] = ctx.stat("swd_warm_items"), ctx.stat("swd_warm_secular_evals"), ctx.stat("swd_exact_secular_evals")
            mk["c0"] = {k: ctx.stat(f"swd_warm_cause_{k}") for k in range(4, 12)}; mk["w0"] = (ctx.stat("swd_warm_walked_chains"), ctx.stat("swd_warm_wide_chains"), ctx.stat("swd_exact_declined_chains"))
            mk["a0"] = (smp_box[0].naccepted_live().sum(), smp_box[0].ntraj_live().sum()) if hasattr(smp_box[0], "naccepted_live") else None
            torch.cuda.synchronize(); mk["t0"] = time.perf_counter()
        if s == W0 + K:
            ctx.check(ctx.L.rfs_synchronize(ctx.h)); torch.cuda.synchronize(); mk["t1"] = time.perf_counter()
            mk["f1"] = ctx.stat("flow_chain_steps"); mk["d1"] = ctx.stat("swd_warm_declined_chains")
            mk["i1"], mk["e1"], mk["x1"] = ctx.stat("swd_warm_items"), ctx.stat("swd_warm_secular_evals"), ctx.stat("swd_exact_secular_evals")
            mk["c1"] = {k: ctx.stat(f"swd_warm_cause_{k}") for k in range(4, 12)}; mk["w1"] = (ctx.stat("swd_warm_walked_chains"), ctx.stat("swd_warm_wide_chains"), ctx.stat("swd_exact_declined_chains"))
            mk["U"] = float(st["Ucur"].median().item())
    smp.sample_flow(x_init=xs, max_steps=W0 + K + 1, step_hook=hook)
    el = mk["t1"] - mk["t0"]
    acc = smp.naccepted.sum() / max(smp.ntrajectories.sum(), 1)
    items = max(mk["i1"] - mk["i0"], 1)
    print(f"dt {dt}: accept ratio {acc:.3f} ({smp.ntrajectories.sum()} trajectories), {el / K * 1e3:.3f} ms per device step, "
          f"{(mk['f1'] - mk['f0']) / el / 1e6:.3f} M evals/s, warm {(mk['e1'] - mk['e0']) / items:.2f} + exact {(mk['x1'] - mk['x0']) / items:.2f} evals/item, "
          f"handed back {(mk['d1'] - mk['d0']) / K:.1f} chains/step, withdrawn {smp.flow_withdrawn}; causes/step "
          f"{ {k: round((mk['c1'][k] - mk['c0'][k]) / K, 1) for k in mk['c0'] if mk['c1'][k] != mk['c0'][k]} }, walked {(mk['w1'][0] - mk['w0'][0]) / K:.0f} wide {(mk['w1'][1] - mk['w0'][1]) / K:.0f} "
          f"exact-declined {(mk['w1'][2] - mk['w0'][2]) / K:.1f} chains/step; U median {mk['U']:.2f}", flush=True)
    joint._ctx.close(); joint._ctx = None
