"""Leapfrog (flow) step time of 8192 chains at one of bench.py's configurations for several settings of the reference-root
stage (swd_warm_exact, swd_exact_group, swd_exact_runup).   python3 scripts/exact_sweep.py [config=1] [steps=40]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rfsurfhmc_amd.model.model_rf import ReceiverFunc
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
cfg = bench.CONFIGS[int(sys.argv[1]) if len(sys.argv) > 1 else 1]
nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 40
n, nt, nchain = cfg["n"], cfg["nt"], 8192
t = np.linspace(5, 44, bench.NPER)
dev = torch.device("cuda")
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
x_true = bench.true_model(n)
bounds = bench.bounds_of(x_true)
xs = np.clip(bench.make_models(nchain, 991206, n), bounds[:, 0], bounds[:, 1])
settings = [(0, 5, 2), (1, 5, 2), (1, 5, 1), (1, 4, 2), (1, 3, 2), (1, 3, 1), (1, 8, 2), (1, 2, 1), (1, 10, 2)]
if len(sys.argv) > 3:
    settings = [tuple(int(v) for v in s.split(",")) for s in sys.argv[3:]]
for exact, G, RU in settings:
    joint = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(bench.RAY_P, nt, cfg["dt"], bench.GAUSS, bench.TSHIFT, bench.WATER, "P", "freq"), SurfWD(tRc=t))
    drf, dswd, flag = joint.forward(x_true); joint.set_obsdata(drf, dswd)
    ctx = joint._ensure(n)
    ctx.set_option("swd_warm_exact", exact); ctx.set_option("swd_exact_group", G); ctx.set_option("swd_exact_runup", RU)
    st = joint.flow_state(tt(xs), torch.full((nchain,), 0.002, dtype=torch.float64, device=dev), tt(bounds))
    st["p"].copy_(tt(0.5 * np.random.default_rng(7).standard_normal(xs.shape))); st["rem"].fill_(1 << 30); st["fresh"].fill_(1)
    for _ in range(12):
        joint.flow_step(st)
    torch.cuda.synchronize(); ctx.L.rfs_synchronize(ctx.h)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(nrep):
            joint.flow_step(st)
        torch.cuda.synchronize(); ctx.L.rfs_synchronize(ctx.h)
        best = min(best, (time.perf_counter() - t0) / nrep)
    it = max(ctx.stat("swd_warm_items"), 1)
    print(f"exact {exact} G {G} runup {RU}: {best * 1e3:.3f} ms per step; exact-stage evals/item {ctx.stat('swd_exact_secular_evals') / it:.2f}, "
          f"warm {ctx.stat('swd_warm_secular_evals') / it:.2f}, handed back {ctx.stat('swd_warm_declined_chains')}", flush=True)
    del joint, st
