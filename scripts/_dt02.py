"""The bench's dt = 0.02 side leg on its own (headline burn-in first), for a kernel trace: what makes its steps 11 ms?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
dt2 = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
cfg = bench.CONFIGS[1]; n = cfg["n"]
dev = torch.device("cuda:0")
joint, x_true, bounds = bench.make_joint(cfg, 0)
barrier = lambda: torch.cuda.synchronize()
rep, xs, el, ev, x_end, mis = bench.sampler_leg(cfg, 1, joint, x_true, bounds, 8192, 0, dev, 200, 300, barrier, kind="hmc", dt=bench.TUNED_DT, mode="reference_roots")
print("headline", rep["ms_per_step"], flush=True)
ctx = joint._ensure(n)
ctx.check(ctx.L.rfs_synchronize(ctx.h)); torch.cuda.synchronize()
for d0 in [float(v) for v in sys.argv[2:]]:
    r0, *_ = bench.sampler_leg(cfg, 1, joint, x_true, bounds, 8192, 0, dev, 100, 60, barrier, kind="hmc", xs=x_end, groups=False, dt=d0, mode="reference_roots")
    print("leg", d0, r0["ms_per_step"], r0["accept_ratio"], flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
r, *_ = bench.sampler_leg(cfg, 1, joint, x_true, bounds, 8192, 0, dev, 100, 60, barrier, kind="hmc", xs=x_end, groups=False, dt=dt2, mode="reference_roots")
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
print("leg", dt2, r["ms_per_step"], r["accept_ratio"], r["root_search"], r.get("root_search_failures"), flush=True)
