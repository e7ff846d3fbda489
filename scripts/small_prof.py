import sys, time, ctypes; sys.path.insert(0, '.')
import numpy as np, torch
from scripts.bench_configs_lib import models
from rfsurfhmc_amd._lib import K_NAMES
from rfsurfhmc_amd.model.model_rf import ReceiverFunc
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
def prof(label, model, x0, n, thk0, vs0, nc):
    out = model.forward(x0)
    if isinstance(model, Joint_RF_SWD): model.set_obsdata(out[0], out[1])
    else: model.set_obsdata(out[0])
    x = torch.from_numpy(models(n, thk0, vs0, nc)).cuda()
    ctx = model._ensure(n)
    for _ in range(3): model.misfit_and_grad_device(x)
    torch.cuda.synchronize(); ctx.check(ctx.L.rfs_enable_timing(ctx.h, 1))
    K = 10
    for _ in range(K): model.misfit_and_grad_device(x)
    ctx.check(ctx.L.rfs_synchronize(ctx.h)); torch.cuda.synchronize()
    ms = np.zeros(len(K_NAMES)); cnt = np.zeros(len(K_NAMES), dtype=np.int32)
    ctx.check(ctx.L.rfs_kernel_ms_sum(ctx.h, ms.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p)))
    ctx.check(ctx.L.rfs_enable_timing(ctx.h, 0))
    print(label, nc, {k: round(ms[i] / max(cnt[i], 1), 3) for i, k in enumerate(K_NAMES)})
thk10 = np.array([3., 3, 4, 5, 5, 6, 7, 8, 10, 0]); vs10 = np.linspace(2.9, 4.6, 10); t36 = np.arange(5., 41.)
for nc in (1, 64):
    prof("cfg1", SurfWD(tRc=t36, tRg=t36), np.hstack((vs10, thk10)), 10, thk10, vs10, nc)
t40 = np.linspace(5, 44, 40)
thk30 = np.full(30, 2.0); thk30[-1] = 0; vs30 = np.linspace(2.8, 4.6, 30)
for nc in (1, 64):
    prof("cfg2", Joint_RF_SWD(1, 1, ReceiverFunc(0.045, 512, 0.1, 1.5, 5.0, 0.001, "P", "freq"), SurfWD(tRc=t40)), np.hstack((vs30, thk30)), 30, thk30, vs30, nc)
