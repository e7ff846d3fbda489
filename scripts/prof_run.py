"""Small fixed workload for rocprofv3: 3 joint evaluations of 8192 chains (config 2)."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
import bench
from rfsurfhmc_amd.model.model_rf import ReceiverFunc
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
nchain = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t = np.linspace(5, 44, 40)
joint = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(0.045, 512, 0.1, 1.5, 5.0, 0.001, "P", "freq"), SurfWD(tRc=t))
drf, dswd, flag = joint.forward(bench.true_model()); joint.set_obsdata(drf, dswd)
x = torch.from_numpy(bench.make_models(nchain, 991206)).cuda()
for _ in range(nrep):
    out = joint.misfit_and_grad_device(x)
torch.cuda.synchronize()
ctx = joint._ensure(30); ctx.L.rfs_synchronize(ctx.h)
print("done", float(out[0].sum()))
