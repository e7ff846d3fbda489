"""Host side of a sampler's continuous-flow run, steps 40..70 under cProfile.
usage: prof_flow_host.py [hmc|da]   (hmc: the bench's 30-layer chains, dt 0.002, L ~ U{5..20};  da: configs[3] shape)"""
import sys, cProfile, pstats, time; sys.path.insert(0, '.')
import numpy as np, torch
import bench
from rfsurfhmc_amd.model.model_rf import ReceiverFunc
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
kind = sys.argv[1] if len(sys.argv) > 1 else "hmc"
S0 = int(sys.argv[2]) if len(sys.argv) > 2 else 40
QUIET = len(sys.argv) > 3
n, nchain = (50 if kind == "da" else 30), 8192
t = np.linspace(5, 44, 40)
joint = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(bench.RAY_P, 512, 0.1, bench.GAUSS, bench.TSHIFT, bench.WATER, "P", "freq"), SurfWD(tRc=t))
x_true = bench.true_model(n)
drf, dswd, flag = joint.forward(x_true); joint.set_obsdata(drf, dswd)
bounds = bench.bounds_of(x_true)
if kind == "da":
    from rfsurfhmc_amd.pyhmc.hmcda import HMCDualAveraging
    rs = np.random.default_rng(3)
    xs = np.clip(x_true[None, :] * (1 + 0.02 * rs.standard_normal((nchain, 2 * n))), bounds[:, 0], bounds[:, 1])
    xs[:, :n] = np.sort(xs[:, :n], axis=1)
    smp = HMCDualAveraging(joint, bounds, 0.1, 10, 10, 0.65, 991206, 100, 20, myrank=0, name="b", outdir=None, nchains=nchain, verbose=False, store_syn=False)
else:
    from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
    xs = bench.make_models(nchain, 991206, n=n)
    smp = HamitonianMC(joint, bounds, 0.002, [5, 20], 10, 991206, 100, 20, myrank=0, name="b", outdir=None, nchains=nchain, verbose=False, store_syn=False)
pr = cProfile.Profile()
def hook(s, st):
    if s == S0: torch.cuda.synchronize(); pr.enable(); hook.t0 = time.perf_counter()
    if s == S0 + 30: torch.cuda.synchronize(); pr.disable(); print("ms/step", (time.perf_counter() - hook.t0) / 30 * 1e3, 'from step', S0)
smp.sample_flow(x_init=xs, max_steps=S0 + 32, step_hook=hook)
if not QUIET: pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
