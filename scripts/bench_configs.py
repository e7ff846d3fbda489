"""Throughput of the other BASELINE.json configurations (parity-test cases, not bench lines):
config 4 shape (50 layers), config 5 shape (nt=2048), config 2 + Rg, SWD-only 10-layer (config 1 shape)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from rfsurfhmc_amd.model.model_rf import ReceiverFunc
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD

def models(n, thk0, vs0, nchain, seed=1):
    lo = np.maximum(vs0 - 0.8 * vs0, 1.5); hi = np.minimum(vs0 + 0.8 * vs0, 5.0)
    rng = np.random.default_rng(seed)
    v = np.sort(lo + (hi - lo) * rng.random((nchain, n)), axis=1)
    h = thk0 * (0.8 + 0.4 * rng.random((nchain, n))); h[:, -1] = 1.0
    return np.hstack((v, h))

def run(label, model, x0, xs, reps=5):
    out = model.forward(x0)
    if isinstance(model, Joint_RF_SWD): model.set_obsdata(out[0], out[1])
    else: model.set_obsdata(out[0])
    x = torch.from_numpy(xs).cuda()
    for _ in range(2): r = model.misfit_and_grad_device(x)
    torch.cuda.synchronize()
    ctx = model._ensure(xs.shape[1] // 2); ctx.L.rfs_synchronize(ctx.h)
    t0 = time.perf_counter()
    for _ in range(reps): r = model.misfit_and_grad_device(x)
    ctx.L.rfs_synchronize(ctx.h); torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    print(f"{label:46s} {xs.shape[0]:6d} chains  {el*1e3:8.2f} ms/step  {xs.shape[0]/el:10.0f} evals/s  fails {int((r[3]==0).sum())}")

nchain = 8192
t40 = np.linspace(5, 44, 40)
thk30 = np.full(30, 2.0); thk30[-1] = 0; vs30 = np.linspace(2.8, 4.6, 30)
thk50 = np.full(50, 1.2); thk50[-1] = 0; vs50 = np.linspace(2.8, 4.6, 50)
rf512 = lambda: ReceiverFunc(0.045, 512, 0.1, 1.5, 5.0, 0.001, "P", "freq")
run("config 2: 30 layers, RF 512 + 40 Rc", Joint_RF_SWD(1, 1, rf512(), SurfWD(tRc=t40)), np.hstack((vs30, thk30)), models(30, thk30, vs30, nchain))
run("config 2 + 40 Rg", Joint_RF_SWD(1, 1, rf512(), SurfWD(tRc=t40, tRg=t40)), np.hstack((vs30, thk30)), models(30, thk30, vs30, nchain))
run("config 4 shape: 50 layers, RF 512 + 40 Rc", Joint_RF_SWD(1, 1, rf512(), SurfWD(tRc=t40)), np.hstack((vs50, thk50)), models(50, thk50, vs50, nchain))
run("config 5: 30 layers, RF 2048 (dt 0.025) + 40 Rc", Joint_RF_SWD(1, 1, ReceiverFunc(0.045, 2048, 0.025, 1.5, 5.0, 0.001, "P", "freq"), SurfWD(tRc=t40)), np.hstack((vs30, thk30)), models(30, thk30, vs30, nchain))
thk10 = np.array([3., 3, 4, 5, 5, 6, 7, 8, 10, 0]); vs10 = np.linspace(2.9, 4.6, 10); t36 = np.arange(5., 41.)
run("config 1 shape: 10 layers, SWD only 36 Rc + 36 Rg", SurfWD(tRc=t36, tRg=t36), np.hstack((vs10, thk10)), models(10, thk10, vs10, nchain))
run("RF only, 30 layers, nt 512", rf512(), np.hstack((vs30, thk30)), models(30, thk30, vs30, nchain))
