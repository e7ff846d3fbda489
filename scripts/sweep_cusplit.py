import sys, time, ctypes; sys.path.insert(0, '.')
import numpy as np, torch
import bench
from rfsurfhmc_amd._lib import K_NAMES
from rfsurfhmc_amd.model.model_rf import ReceiverFunc
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
nchain = 8192
t = np.linspace(5, 44, 40)
joint = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(0.045, 512, 0.1, 1.5, 5.0, 0.001, "P", "freq"), SurfWD(tRc=t))
drf, dswd, flag = joint.forward(bench.true_model()); joint.set_obsdata(drf, dswd)
x = torch.from_numpy(bench.make_models(nchain, 991206)).cuda()
ctx = joint._ensure(30)
ref = None
for mode in (0, 1, 2, 0, 1, 2):
    ctx.check(ctx.L.rfs_set_option(ctx.h, b"cu_split", mode))
    for _ in range(3): out = joint.misfit_and_grad_device(x)
    torch.cuda.synchronize(); ctx.L.rfs_synchronize(ctx.h)
    ctx.L.rfs_enable_timing(ctx.h, 1)
    t0 = time.perf_counter()
    for _ in range(10): out = joint.misfit_and_grad_device(x)
    ctx.L.rfs_synchronize(ctx.h); torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 10
    ms = np.zeros(7); cnt = np.zeros(7, dtype=np.int32)
    ctx.L.rfs_kernel_ms_sum(ctx.h, ms.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p))
    ctx.L.rfs_enable_timing(ctx.h, 0)
    g = out[1].cpu().numpy()
    if ref is None: ref = g
    print(f"cu_split={mode} step {el*1e3:7.2f} ms  {nchain/el:10.0f} evals/s  same grad {np.array_equal(g, ref)}  " +
          " ".join(f"{k}={ms[i]/max(cnt[i],1):.2f}" for i, k in enumerate(K_NAMES)))
