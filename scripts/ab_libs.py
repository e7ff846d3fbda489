"""A/B of two builds of the library on ONE box: alternating runs of the bench workload (bare evaluation loop and the
leapfrog flow step), each build in its own process.  usage: python scripts/ab_libs.py libA.so libB.so [reps]"""
import subprocess, sys, json, os
libs = sys.argv[1:3]; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
code = r'''
import sys, time, ctypes; sys.path.insert(0, '.')
import numpy as np, torch, bench
from rfsurfhmc_amd._lib import K_NAMES
from rfsurfhmc_amd.model.model_rf import ReceiverFunc
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
t = np.linspace(5, 44, 40)
joint = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(bench.RAY_P, 512, 0.1, bench.GAUSS, bench.TSHIFT, bench.WATER, "P", "freq"), SurfWD(tRc=t))
drf, dswd, flag = joint.forward(bench.true_model()); joint.set_obsdata(drf, dswd)
x = torch.from_numpy(bench.make_models(8192, 991206)).cuda()
ctx = joint._ensure(30)
for _ in range(32): out = joint.misfit_and_grad_device(x)
torch.cuda.synchronize(); ctx.L.rfs_synchronize(ctx.h)
ctx.L.rfs_enable_timing(ctx.h, 2 * (1 << 4))
t0 = time.perf_counter()
for _ in range(20): out = joint.misfit_and_grad_device(x)
ctx.L.rfs_synchronize(ctx.h); torch.cuda.synchronize()
el = (time.perf_counter() - t0) / 20
ms = np.zeros(len(K_NAMES)); cnt = np.zeros(len(K_NAMES), dtype=np.int32)
ctx.L.rfs_kernel_ms_sum(ctx.h, ms.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p))
print("RESULT %.3f %.3f %.9e" % (el * 1e3, ms[4] / max(cnt[4], 1), float(out[0].sum())))
'''
for r in range(reps):
    for lib in libs:
        env = dict(os.environ, RFSURF_LIB=os.path.abspath(lib))
        o = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        line = [l for l in o.stdout.splitlines() if l.startswith("RESULT")]
        print(os.path.basename(lib), line[0] if line else o.stderr[-300:], " ".join(l for l in o.stderr.splitlines() if "[rfs]" in l), flush=True)
