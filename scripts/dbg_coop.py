import sys, ctypes; sys.path.insert(0, '.')
import numpy as np, torch, bench
from rfsurfhmc_amd.model.model_surf import SurfWD
t = np.linspace(5, 44, 40); swd = SurfWD(tRc=t)
d, f = swd.forward(bench.true_model()); swd.set_obsdata(d)
x = torch.from_numpy(bench.make_models(8192, 991206)).cuda()
for _ in range(3): out = swd.misfit_and_grad_device(x)
torch.cuda.synchronize()
ctx = swd._ensure(30)
buf = np.zeros((128, 8), dtype=np.int64)
ctx.L.rfs_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
ctx.L.rfs_debug_read(ctx.h, buf.ctypes.data_as(ctypes.c_void_p), buf.size)
m = buf.mean(axis=0)
print("per block mean (cycles): req %.3g  barrier-wait %.3g  apply %.3g  advance %.3g  evals %.0f | producer total %.3g of which barrier-wait %.3g" % tuple(m[:7]))
print("per eval: req %.0f bar %.0f apply %.0f adv %.0f | producer busy %.0f" % (m[0]/m[4], m[1]/m[4], m[2]/m[4], m[3]/m[4], (m[5]-m[6])/m[4]))
