"""Step time of config 2 with and without the per-group HIP-event timing of bench.py (measurement overhead)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
import bench
from rfsurfhmc_amd.model.model_rf import ReceiverFunc
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
t = np.linspace(5, 44, bench.NPER)
joint = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(bench.RAY_P, bench.NT, bench.DT, bench.GAUSS, bench.TSHIFT, bench.WATER, "P", "freq"), SurfWD(tRc=t))
drf, dswd, flag = joint.forward(bench.true_model()); joint.set_obsdata(drf, dswd)
x = torch.from_numpy(bench.make_models(8192, 991206)).cuda()
ctx = joint._ensure(bench.N_LAYER)
for timing in (0, 1, 0, 1):
    ctx.check(ctx.L.rfs_enable_timing(ctx.h, timing))
    for _ in range(3): joint.misfit_and_grad_device(x)
    torch.cuda.synchronize(); ctx.check(ctx.L.rfs_synchronize(ctx.h))
    t0 = time.perf_counter()
    for _ in range(20): joint.misfit_and_grad_device(x)
    ctx.check(ctx.L.rfs_synchronize(ctx.h)); torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 20
    print("timing", timing, round(el * 1e3, 3), "ms/step", round(8192 / el), "evals/s")
