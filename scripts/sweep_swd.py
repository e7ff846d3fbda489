import sys, time, ctypes; sys.path.insert(0, '.')
import numpy as np, torch
import bench
from rfsurfhmc_amd._lib import K_NAMES
from rfsurfhmc_amd.model.model_surf import SurfWD
nchain = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
t = np.linspace(5, 44, 40)
swd = SurfWD(tRc=t)
x0 = bench.true_model()
d, f = swd.forward(x0); swd.set_obsdata(d)
x = torch.from_numpy(bench.make_models(nchain, 991206)).cuda()
ref = None
for G in [int(a) for a in sys.argv[2:]] or (1, 8):
    ctx = swd._ensure(30)
    ctx.check(ctx.L.rfs_set_option(ctx.h, b"swd_lanes_per_chain", G))
    for _ in range(2): out = swd.misfit_and_grad_device(x)
    torch.cuda.synchronize(); ctx.L.rfs_synchronize(ctx.h)
    ctx.L.rfs_enable_timing(ctx.h, 1)
    for _ in range(5): out = swd.misfit_and_grad_device(x)
    ms = np.zeros(7); cnt = np.zeros(7, dtype=np.int32)
    ctx.L.rfs_kernel_ms_sum(ctx.h, ms.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p))
    ctx.L.rfs_enable_timing(ctx.h, 0)
    dd = out[2].cpu().numpy()
    if ref is None: ref = dd
    print(f"G={G:2d} roots {ms[4]/cnt[4]:7.2f} ms eigen {ms[5]/cnt[5]:.2f}  c dev vs first: max rel {np.abs(dd-ref).max()/np.abs(ref).max():.2e} nonexact {(dd!=ref).sum()}/{dd.size}")
