import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
cfg = bench.CONFIGS[1]; n = cfg["n"]; nchain = 8192
joint, x_true, bounds = bench.make_joint(cfg, 0)
ctx = joint._ensure(n)
xs = bench.make_models(nchain, 991206, n)
smp = HamitonianMC(joint, bounds, 0.002, [5, 20], 10, 991206, 100, 20, myrank=0, name="bench", outdir=None, nchains=nchain, verbose=False, store_syn=False)
last = [0]
def hook(s, st):
    d = ctx.stat("swd_warm_declined_chains")
    done = st["done"].cpu().numpy()
    if 20 <= s < 50:
        print(s, "declined", d - last[0], "done1", int((done == 1).sum()), "rej", int((done == 2).sum()), "acc", int((done == 3).sum()),
              "fresh", int(st["fresh"].sum()), "ok0", int((st["ok"] == 0).sum()))
    last[0] = d
smp.sample_flow(x_init=xs, max_steps=50, step_hook=hook)
