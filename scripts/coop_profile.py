"""Where a round of the cooperative root search goes: cycle stamps of the consumer wave (profiling build of the library,
-DRFS_COOP_PROFILE -> ab/librfsurf_prof.so; run with RFSURF_LIB=ab/librfsurf_prof.so)."""
import sys, ctypes; sys.path.insert(0, '.')
import numpy as np, torch
import bench
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.model.model_rf import ReceiverFunc
from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
nchain = 8192
t = np.linspace(5, 44, bench.NPER)
joint = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(bench.RAY_P, 512, 0.1, bench.GAUSS, bench.TSHIFT, bench.WATER, "P", "freq"), SurfWD(tRc=t))
drf, dswd, flag = joint.forward(bench.true_model()); joint.set_obsdata(drf, dswd)
x = torch.from_numpy(bench.make_models(nchain, 991206)).cuda()
ctx = joint._ensure(30)
ctx.L.rfs_debug_coop_profile.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
for shape, pcu in [(81, 1)]:
    ctx.check(ctx.L.rfs_set_option(ctx.h, b"swd_coop_shape", shape))
    ctx.check(ctx.L.rfs_set_option(ctx.h, b"swd_coop_blocks_per_cu", pcu))
    for _ in range(3): joint.misfit_and_grad_device(x)
    torch.cuda.synchronize()
    nb = 128 if shape == 81 else 256
    buf = np.zeros(16 * 1024, dtype=np.int64)
    ctx.check(ctx.L.rfs_debug_coop_profile(ctx.h, buf.ctypes.data_as(ctypes.c_void_p), 16 * 1024))
    b = buf.reshape(1024, 16)[:nb]
    r = b[:, 0].astype(float)
    names = ["rounds", "request", "halfspace+deepest", "B0->last chunk ready", "last-chunk apply", "state machine",
             "  sm: dispatch", "  sm: looptop", "  sm: a1 (Neville)", "  sm: finish+fail", "  sm: half+scan", "  sm: new period"]
    print(f"shape {shape}: per-round cycles of the consumer (mean over {nb} blocks; min..max of block means)")
    for i, nm in enumerate(names):
        v = b[:, i] / (r if i else 1)
        print(f"  {nm:24s} {v.mean():10.0f}   {v.min():10.0f} .. {v.max():10.0f}")
    tot = (b[:, 1] + b[:, 3] + b[:, 4] + b[:, 5]) / r
    print(f"  {'sum per round':24s} {tot.mean():10.0f}; rounds x sum = {np.mean(r * tot) / 1e6:.2f} Mcycles (max block {np.max(r*tot)/1e6:.2f})")
