import sys, time; sys.path.insert(0, '.')
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
ts = []; bad = []
def hook(s, st):
    torch.cuda.synchronize(); ts.append(time.perf_counter()); bad.append(int((st["ok"] == 0).sum().item()))
smp.sample_flow(x_init=xs, max_steps=70, step_hook=hook)
d = np.diff(ts) * 1e3
print("per-step ms (synchronised before every step):", np.round(d, 1).tolist())
print("chains with ok = 0 after each step:", bad)
# bare evaluation of the current models for comparison
x = torch.from_numpy(xs).cuda()
for _ in range(2): joint.misfit_and_grad_device(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): joint.misfit_and_grad_device(x)
torch.cuda.synchronize(); print("bare eval ms", (time.perf_counter() - t0) / 5 * 1e3)
