"""cProfile of the host side of HMCDualAveraging.sample_flow at the configs[3] shape (8192 chains x 50 layers)."""
import sys, cProfile, pstats; sys.path.insert(0, '.')
import numpy as np, torch
import bench
from rfsurfhmc_amd.model.model_rf import ReceiverFunc
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
from rfsurfhmc_amd.pyhmc.hmcda import HMCDualAveraging
n, nchain = 50, 8192
t = np.linspace(5, 44, 40)
joint = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(bench.RAY_P, 512, 0.1, bench.GAUSS, bench.TSHIFT, bench.WATER, "P", "freq"), SurfWD(tRc=t))
x_true = bench.true_model(n)
drf, dswd, flag = joint.forward(x_true); joint.set_obsdata(drf, dswd)
bounds = bench.bounds_of(x_true)
rs = np.random.default_rng(3)
xs = np.clip(x_true[None, :] * (1 + 0.02 * rs.standard_normal((nchain, 2 * n))), bounds[:, 0], bounds[:, 1])
xs[:, :n] = np.sort(xs[:, :n], axis=1)
smp = HMCDualAveraging(joint, bounds, 0.1, 10, 10, 0.65, 991206, 100, 20, myrank=0, name="b", outdir=None, nchains=nchain, verbose=False, store_syn=False)
pr = cProfile.Profile()
import time
def hook(s, st):
    if s == 40: pr.enable(); hook.t0 = time.perf_counter()
    if s == 70: pr.disable(); print("ms/step", (time.perf_counter() - hook.t0) / 30 * 1e3)
smp.sample_flow(x_init=xs, max_steps=72, step_hook=hook)
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
