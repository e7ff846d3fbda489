"""Latency of one joint evaluation at small chain counts, with the per-group kernel times (HIP events)."""
import sys, time, ctypes; sys.path.insert(0, '.')
import numpy as np, torch, bench
from rfsurfhmc_amd._lib import K_NAMES
from rfsurfhmc_amd.model.model_rf import ReceiverFunc
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
t = np.linspace(5, 44, 40)
joint = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(bench.RAY_P, 512, 0.1, bench.GAUSS, bench.TSHIFT, bench.WATER, "P", "freq"), SurfWD(tRc=t))
drf, dswd, flag = joint.forward(bench.true_model()); joint.set_obsdata(drf, dswd)
ctx = joint._ensure(30)
import os
for kv in filter(None, os.environ.get("RFS_OPTS", "").split(",")):      # e.g. RFS_OPTS=swd_speculate=2,swd_segments=4
    k, v = kv.split("="); ctx.check(ctx.L.rfs_set_option(ctx.h, k.encode(), int(v))); print("option", k, v)
for nchain in [int(a) for a in sys.argv[1:]] or [1, 8, 64, 512, 2048]:
    x = torch.from_numpy(bench.make_models(nchain, 991206)).cuda()
    for _ in range(36): joint.misfit_and_grad_device(x)
    ctx.check(ctx.L.rfs_synchronize(ctx.h)); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): joint.misfit_and_grad_device(x)
    ctx.check(ctx.L.rfs_synchronize(ctx.h)); torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 20 * 1e3
    ctx.check(ctx.L.rfs_enable_timing(ctx.h, 1))
    for _ in range(10): joint.misfit_and_grad_device(x)
    ctx.check(ctx.L.rfs_synchronize(ctx.h)); torch.cuda.synchronize()
    ms = np.zeros(len(K_NAMES)); cnt = np.zeros(len(K_NAMES), dtype=np.int32)
    ctx.check(ctx.L.rfs_kernel_ms_sum(ctx.h, ms.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p)))
    ctx.check(ctx.L.rfs_enable_timing(ctx.h, 0))
    print(f"{nchain:6d} chains: {wall:7.3f} ms/eval  " + "  ".join(f"{k} {ms[i] / max(cnt[i], 1):.3f}" for i, k in enumerate(K_NAMES)), flush=True)
