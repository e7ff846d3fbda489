"""A/B helper: step time and root-search kernel time of config 2 with an alternative build of the library.
usage: python scripts/ab_roots.py [path/to/other/librfsurf_hip.so]"""
import sys, time, ctypes; sys.path.insert(0, '.')
import numpy as np, torch
import rfsurfhmc_amd._lib as L
if len(sys.argv) > 1:
    L.LIBPATH = sys.argv[1]
import bench
from rfsurfhmc_amd._lib import K_NAMES
from rfsurfhmc_amd.model.model_rf import ReceiverFunc
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
t = np.linspace(5, 44, bench.NPER)
joint = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(bench.RAY_P, bench.NT, bench.DT, bench.GAUSS, bench.TSHIFT, bench.WATER, "P", "freq"), SurfWD(tRc=t))
drf, dswd, flag = joint.forward(bench.true_model()); joint.set_obsdata(drf, dswd)
x = torch.from_numpy(bench.make_models(8192, 991206)).cuda()
ctx = joint._ensure(bench.N_LAYER)
for _ in range(3): joint.misfit_and_grad_device(x)
torch.cuda.synchronize(); ctx.check(ctx.L.rfs_synchronize(ctx.h))
ctx.check(ctx.L.rfs_enable_timing(ctx.h, 2 * (1 << K_NAMES.index("swd_roots"))))
t0 = time.perf_counter(); K = 30
for _ in range(K): joint.misfit_and_grad_device(x)
ctx.check(ctx.L.rfs_synchronize(ctx.h)); torch.cuda.synchronize(); el = (time.perf_counter() - t0) / K
ms = np.zeros(len(K_NAMES)); cnt = np.zeros(len(K_NAMES), dtype=np.int32)
ctx.check(ctx.L.rfs_kernel_ms_sum(ctx.h, ms.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p)))
i = K_NAMES.index("swd_roots")
print(sys.argv[1] if len(sys.argv) > 1 else "in-tree", "step %.3f ms  roots %.3f ms" % (el * 1e3, ms[i] / max(cnt[i], 1)))
