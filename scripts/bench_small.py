"""Latency of one evaluation for small chain counts (a user who runs the reference's 1-chain-per-rank set-up):
config-1 shape (10 layers, SWD only, 36 Rc + 36 Rg) and config-2 shape (30 layers, RF 512 + 40 Rc)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from scripts.bench_configs_lib import models
from rfsurfhmc_amd.model.model_rf import ReceiverFunc
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD


def run(label, model, x0, n, thk0, vs0, counts, reps=10):
    out = model.forward(x0)
    if isinstance(model, Joint_RF_SWD): model.set_obsdata(out[0], out[1])
    else: model.set_obsdata(out[0])
    for nc in counts:
        xs = models(n, thk0, vs0, nc)
        # host entry (numpy in, numpy out): what the reference-style sampler calls
        for _ in range(2): model.misfit_and_grad(xs)
        t0 = time.perf_counter()
        for _ in range(reps): model.misfit_and_grad(xs)
        host = (time.perf_counter() - t0) / reps
        x = torch.from_numpy(xs).cuda()
        for _ in range(2): model.misfit_and_grad_device(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): model.misfit_and_grad_device(x)
        torch.cuda.synchronize()
        dev = (time.perf_counter() - t0) / reps
        print(f"{label:40s} {nc:5d} chains  host entry {host*1e3:7.2f} ms ({nc/host:9.0f} evals/s)   device entry {dev*1e3:7.2f} ms ({nc/dev:9.0f} evals/s)")


counts = [int(a) for a in sys.argv[1:]] or [1, 8, 64, 512]
thk10 = np.array([3., 3, 4, 5, 5, 6, 7, 8, 10, 0]); vs10 = np.linspace(2.9, 4.6, 10); t36 = np.arange(5., 41.)
run("config 1 shape (SWD only, 10 layers)", SurfWD(tRc=t36, tRg=t36), np.hstack((vs10, thk10)), 10, thk10, vs10, counts)
t40 = np.linspace(5, 44, 40)
thk30 = np.full(30, 2.0); thk30[-1] = 0; vs30 = np.linspace(2.8, 4.6, 30)
run("config 2 shape (RF 512 + 40 Rc, 30 layers)", Joint_RF_SWD(1, 1, ReceiverFunc(0.045, 512, 0.1, 1.5, 5.0, 0.001, "P", "freq"), SurfWD(tRc=t40)),
    np.hstack((vs30, thk30)), 30, thk30, vs30, counts)
