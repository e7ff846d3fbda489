import ctypes, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from rfsurfhmc_amd.model.model_rf import ReceiverFunc
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
nchain, nsteps, dt = 8192, 6, 0.002
cfg = bench.CONFIGS[1]; n, nt = cfg["n"], cfg["nt"]
dev = torch.device("cuda", 0); t = np.linspace(5, 44, bench.NPER)
def make():
    j = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(bench.RAY_P, nt, cfg["dt"], bench.GAUSS, bench.TSHIFT, bench.WATER, "P", "freq"), SurfWD(tRc=t))
    x_true = bench.true_model(n); drf, dswd, flag = j.forward(x_true); j.set_obsdata(drf, dswd)
    return j, x_true
je, x_true = make(); ctxe = je._ensure(n); ctxe.set_option("swd_warm_start", 0)
jw2, _ = make(); ctx2 = jw2._ensure(n); ctx2.set_option("swd_warm_start", 2)
bounds = bench.bounds_of(x_true)
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
xs = bench.make_models(nchain, seed=991206, n=n)
rng = np.random.default_rng(7)
x = tt(xs).clone(); p = tt(0.5 * rng.standard_normal(xs.shape)); lo, hi = tt(bounds[:, 0]), tt(bounds[:, 1])
xprev = None
for s in range(nsteps):
    d0 = ctx2.stat("swd_warm_declined_chains")
    mw, gw, dw, fw = jw2.misfit_and_grad_device(x)
    me, ge, de, fe = je.misfit_and_grad_device(x)
    torch.cuda.synchronize()
    gr = ((gw - ge).abs().amax(dim=1) / ge.abs().amax(dim=1))
    rr = ((dw[:, nt:] - de[:, nt:]).abs() / de[:, nt:]).amax(dim=1)
    w = int(gr.argmax())
    mv = (x - xprev).abs().amax(dim=1) if xprev is not None else torch.zeros(nchain, device=dev)
    print(f"step {s}: declined this step {ctx2.stat('swd_warm_declined_chains') - d0}; grad rel: median {gr.median():.2e} p99 {gr.quantile(0.99):.2e} max {gr.max():.2e} "
          f"at chain {w} (root rel there {rr[w]:.2e}, misfit w {mw[w]:.6e} e {me[w]:.6e}, |g|max {ge[w].abs().max():.3e}, move {mv[w]:.2e}); "
          f"moves: median {mv.median():.2e} max {mv.max():.2e}; |g| median {ge.abs().amax(dim=1).median():.2e}")
    big = (gr > 1e-4).nonzero().flatten()[:5].tolist()
    for c in big:
        j = int((gw[c] - ge[c]).abs().argmax())
        print("   chain", c, "comp", j, "gw", gw[c, j].item(), "ge", ge[c, j].item(), "rootrel", rr[c].item(), "rf part same?",
              "dsyn rf maxdiff", (dw[c, :nt] - de[c, :nt]).abs().max().item())
    xprev = x.clone()
    p = p - dt * gw
    x = x + dt * p
    over, under = x > hi, x < lo
    x = torch.where(over, 2 * hi - x, x); x = torch.where(under, 2 * lo - x, x)
    p = torch.where(over | under, -p, p)
