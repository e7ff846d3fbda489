"""Step time against the number of periods whose eigenfunction kernels run early on the RF half (same box, one process)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch, bench
from rfsurfhmc_amd.model.model_rf import ReceiverFunc
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
t = np.linspace(5, 44, 40)
joint = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(bench.RAY_P, 512, 0.1, bench.GAUSS, bench.TSHIFT, bench.WATER, "P", "freq"), SurfWD(tRc=t))
drf, dswd, flag = joint.forward(bench.true_model()); joint.set_obsdata(drf, dswd)
x = torch.from_numpy(bench.make_models(8192, 991206)).cuda()
ctx = joint._ensure(30)
for _ in range(32): joint.misfit_and_grad_device(x)
for rep in range(2):
    for k in (0, 8, 12, 16, 20, 24, 28, 32):
        ctx.check(ctx.L.rfs_set_option(ctx.h, b"early_eigen_periods", k))
        for _ in range(3): joint.misfit_and_grad_device(x)
        torch.cuda.synchronize(); ctx.L.rfs_synchronize(ctx.h)
        t0 = time.perf_counter()
        for _ in range(15): joint.misfit_and_grad_device(x)
        ctx.L.rfs_synchronize(ctx.h); torch.cuda.synchronize()
        print(f"early {k:2d}: {(time.perf_counter() - t0) / 15 * 1e3:.3f} ms/eval", flush=True)
