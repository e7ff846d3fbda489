"""Like dt_sweep.py, but every step size continues the SAME burned-in chains (300 device steps at dt 0.05 first)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dts = [float(v) for v in sys.argv[2:]] or [0.002, 0.02, 0.05, 0.1]
cfg = bench.CONFIGS[1]; nchain = 8192; n = cfg["n"]
joint, x_true, bounds = bench.make_joint(cfg, 0)
ctx = joint._ensure(n)
xs = bench.make_models(nchain, 991206, n)
keep = {}
smp = HamitonianMC(joint, bounds, 0.05, [5, 20], 10, 991206, 200, 20, myrank=0, name="b", outdir=None, nchains=nchain, verbose=False, store_syn=False)
smp.sample_flow(x_init=xs, max_steps=301, step_hook=lambda s, st: keep.__setitem__("x", st["x"].clone()) if s == 300 else None)
xb = keep["x"].cpu().numpy()
print("burned in: U median", float(np.median(joint.misfit_and_grad(xb[:512])[0])))
W0 = 40
for dt in dts:
    smp = HamitonianMC(joint, bounds, dt, [5, 20], 10, 991206, 200, 20, myrank=0, name="b", outdir=None, nchains=nchain, verbose=False, store_syn=False)
    mk = {}
    NAMES = ["swd_warm_declined_chains", "swd_warm_items", "swd_warm_secular_evals", "swd_exact_secular_evals", "swd_warm_walked_chains", "swd_warm_wide_chains", "flow_chain_steps"] + [f"swd_warm_cause_{k}" for k in range(4, 12)]
    def hook(s, st):
        if s == W0:
            mk["s0"] = {k: ctx.stat(k) for k in NAMES}; torch.cuda.synchronize(); mk["t0"] = time.perf_counter()
        if s == W0 + K:
            ctx.check(ctx.L.rfs_synchronize(ctx.h)); torch.cuda.synchronize(); mk["t1"] = time.perf_counter(); mk["s1"] = {k: ctx.stat(k) for k in NAMES}
    smp.sample_flow(x_init=xb, max_steps=W0 + K + 1, step_hook=hook)
    d = {k: (mk["s1"][k] - mk["s0"][k]) / K for k in NAMES}
    el = mk["t1"] - mk["t0"]
    print(f"dt {dt}: {el / K * 1e3:.3f} ms/step, accept {smp.naccepted.sum() / max(smp.ntrajectories.sum(), 1):.3f}, evals/item warm {d['swd_warm_secular_evals'] / max(d['swd_warm_items'], 1):.2f} exact {d['swd_exact_secular_evals'] / max(d['swd_warm_items'], 1):.2f}; per step: handed back {d['swd_warm_declined_chains']:.1f}, "
          f"walked {d['swd_warm_walked_chains']:.0f}, wide {d['swd_warm_wide_chains']:.0f}, causes { {k[-2:].strip('_'): round(v, 1) for k, v in d.items() if 'cause' in k and v > 0} }", flush=True)
