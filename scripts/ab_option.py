"""One rfs_set_option knob, values alternating inside ONE process / one box.  usage: ab_option.py <name> <v0> <v1> ..."""
import sys, time, ctypes; sys.path.insert(0, '.')
import numpy as np, torch, bench
from rfsurfhmc_amd._lib import K_NAMES
from rfsurfhmc_amd.model.model_rf import ReceiverFunc
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
name = sys.argv[1].encode(); values = [int(a) for a in sys.argv[2:]]
t = np.linspace(5, 44, 40)
joint = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(bench.RAY_P, 512, 0.1, bench.GAUSS, bench.TSHIFT, bench.WATER, "P", "freq"), SurfWD(tRc=t))
drf, dswd, flag = joint.forward(bench.true_model()); joint.set_obsdata(drf, dswd)
x = torch.from_numpy(bench.make_models(8192, 991206)).cuda()
ctx = joint._ensure(30)
for rep in range(3):
    for v in values:
        ctx.check(ctx.L.rfs_set_option(ctx.h, name, v))
        for _ in range(34): out = joint.misfit_and_grad_device(x)          # lets the schedule settle again
        torch.cuda.synchronize(); ctx.L.rfs_synchronize(ctx.h)
        t0 = time.perf_counter()
        for _ in range(20): out = joint.misfit_and_grad_device(x)
        ctx.L.rfs_synchronize(ctx.h); torch.cuda.synchronize()
        print(f"{name.decode()} = {v}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/eval  checksum {float(out[0].sum()):.9e}", flush=True)
