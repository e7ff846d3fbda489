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
m = buf.mean(axis=0); ne = m[7]
print("per eval cycles: request %.0f | barrier waits %.0f | halfspace+own layer %.0f | apply+LDS %.0f | advance %.0f | evals %.0f | sum %.0f" % (m[0]/ne, m[1]/ne, m[2]/ne, m[3]/ne, m[4]/ne, ne, m[:5].sum()/ne))
