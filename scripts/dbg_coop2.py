import sys, ctypes, numpy as np
sys.path.insert(0, '.')
import torch
from bench import true_model, make_models, N_LAYER, NT, DT, NPER, RAY_P, GAUSS, TSHIFT, WATER
from rfsurfhmc_amd.model.model_rf import ReceiverFunc
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
t = np.linspace(5, 44, NPER)
joint = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(RAY_P, NT, DT, GAUSS, TSHIFT, WATER, "P", "freq"), SurfWD(tRc=t))
drf, dswd, flag = joint.forward(true_model()); joint.set_obsdata(drf, dswd)
x = torch.from_numpy(make_models(8192, 991206)).cuda()
for _ in range(3): out = joint.misfit_and_grad_device(x)
torch.cuda.synchronize()
ctx = joint._ensure(N_LAYER); ctx.L.rfs_synchronize(ctx.h)
d = np.zeros(64, dtype=np.int64)
L = ctypes.CDLL('rfsurfhmc_amd/librfsurf_hip.so')
print(L.rfs_debug_read(d.ctypes.data_as(ctypes.c_void_p)))
nr = d[6]
print("rounds", nr)
print("consumer per round: req %d b0wait %d own %d chunkwait %d apply %d adv %d  total %d" % tuple(list(d[:6] // nr) + [d[:6].sum() // nr]))
for p in range(7):
    print("producer", p, "per round: b0wait %d compute %d chunkwait %d" % tuple(d[8 + 3 * p: 11 + 3 * p] // nr))
