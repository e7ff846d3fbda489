"""Host-pointer entry (rfs_joint_misfit_grad: H2D copy of x, D2H copy of misfit/grad/dsyn/flag) vs the
device-pointer entry, config 2, 8192 chains -- the PCIe-inclusive rate quoted in DESIGN.md."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
import bench
from rfsurfhmc_amd.model.model_rf import ReceiverFunc
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
nchain = 8192
t = np.linspace(5, 44, 40)
joint = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(0.045, 512, 0.1, 1.5, 5.0, 0.001, "P", "freq"), SurfWD(tRc=t))
drf, dswd, flag = joint.forward(bench.true_model()); joint.set_obsdata(drf, dswd)
xs = bench.make_models(nchain, 991206)
ctx = joint._ensure(30)
for split in (1, 0, 1):
    ctx.check(ctx.L.rfs_set_option(ctx.h, b"cu_split", split))
    for _ in range(2): joint.misfit_and_grad(xs)
    t0 = time.perf_counter()
    for _ in range(10): joint.misfit_and_grad(xs)
    el = (time.perf_counter() - t0) / 10
    print(f"cu_split={split} host-pointer entry: {el*1e3:.2f} ms/step  {nchain/el:.0f} evals/s (PCIe + numpy allocation inclusive)")
x = torch.from_numpy(xs).cuda()
for _ in range(2): joint.misfit_and_grad_device(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): joint.misfit_and_grad_device(x)
ctx.L.rfs_synchronize(ctx.h); torch.cuda.synchronize()
el = (time.perf_counter() - t0) / 10
print(f"device-pointer entry: {el*1e3:.2f} ms/step  {nchain/el:.0f} evals/s")
