"""Throughput of the joint misfit+gradient evaluation with the time-domain RF (method="time")."""
import sys, time, ctypes
import numpy as np
sys.path.insert(0, '.')
import torch
from bench import true_model, make_models, N_LAYER, NT, DT, NPER, RAY_P, GAUSS, TSHIFT, WATER
from rfsurfhmc_amd._lib import K_NAMES
from rfsurfhmc_amd.model.model_rf import ReceiverFunc
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
nchain = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
t = np.linspace(5, 44, NPER)
joint = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(RAY_P, NT, DT, GAUSS, TSHIFT, WATER, "P", "time"), SurfWD(tRc=t))
drf, dswd, flag = joint.forward(true_model()); assert flag
joint.set_obsdata(drf, dswd)
x = torch.from_numpy(make_models(nchain, 991206)).cuda()
ctx = joint._ensure(N_LAYER)
for _ in range(2): out = joint.misfit_and_grad_device(x)
torch.cuda.synchronize(); ctx.check(ctx.L.rfs_synchronize(ctx.h)); ctx.check(ctx.L.rfs_enable_timing(ctx.h, 1))
t0 = time.perf_counter(); K = 5
for _ in range(K): out = joint.misfit_and_grad_device(x)
ctx.check(ctx.L.rfs_synchronize(ctx.h)); torch.cuda.synchronize(); el = time.perf_counter() - t0
ms = np.zeros(len(K_NAMES)); cnt = np.zeros(len(K_NAMES), dtype=np.int32)
ctx.check(ctx.L.rfs_kernel_ms_sum(ctx.h, ms.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p)))
print("time-domain joint:", nchain * K / el, "evals/s", el / K * 1e3, "ms/step",
      {k: round(ms[i] / max(cnt[i], 1), 3) for i, k in enumerate(K_NAMES)})
print("misfit finite:", bool(torch.isfinite(out[0]).all()), "flags", int(out[3].sum()))
