"""Kernel experiments inside ONE process / one box (rfs_set_option "experiment"): alternating, several rounds -- the only
comparison that can be trusted (separate gpurun calls land on boxes that differ by several per cent)."""
import sys, time, ctypes; sys.path.insert(0, '.')
import numpy as np, torch, bench
from rfsurfhmc_amd._lib import K_NAMES
from rfsurfhmc_amd.model.model_rf import ReceiverFunc
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
variants = [int(a) for a in sys.argv[1:]] or [0, 1]
t = np.linspace(5, 44, 40)
joint = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(bench.RAY_P, 512, 0.1, bench.GAUSS, bench.TSHIFT, bench.WATER, "P", "freq"), SurfWD(tRc=t))
drf, dswd, flag = joint.forward(bench.true_model()); joint.set_obsdata(drf, dswd)
x = torch.from_numpy(bench.make_models(8192, 991206)).cuda()
ctx = joint._ensure(30)
ctx.check(ctx.L.rfs_set_option(ctx.h, b"early_eigen_periods", 24))
for _ in range(40): joint.misfit_and_grad_device(x)
for rep in range(3):
    for v in variants:
        ctx.check(ctx.L.rfs_set_option(ctx.h, b"experiment", v))
        for _ in range(4): out = joint.misfit_and_grad_device(x)
        torch.cuda.synchronize(); ctx.L.rfs_synchronize(ctx.h)
        ctx.L.rfs_enable_timing(ctx.h, 1)
        t0 = time.perf_counter()
        for _ in range(20): out = joint.misfit_and_grad_device(x)
        ctx.L.rfs_synchronize(ctx.h); torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / 20
        ms = np.zeros(len(K_NAMES)); cnt = np.zeros(len(K_NAMES), dtype=np.int32)
        ctx.L.rfs_kernel_ms_sum(ctx.h, ms.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p))
        ctx.L.rfs_enable_timing(ctx.h, 0)
        per = " ".join(f"{k}={ms[i]/max(cnt[i],1):.3f}" for i, k in enumerate(K_NAMES))
        print(f"variant {v}: {el*1e3:.3f} ms/eval  {per}  checksum {float(out[0].sum()):.9e}", flush=True)
