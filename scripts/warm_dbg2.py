import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from rfsurfhmc_amd.model.model_surf import SurfWD
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_warm import _leapfrog_move
rng = np.random.default_rng(4)
n, nchain = 12, 256
thk = np.r_[np.full(n - 1, 3.0), 0.0]; vs = np.linspace(2.8, 4.5, n); x0 = np.hstack((vs, thk))
t = np.linspace(6.0, 36.0, 7); dev = torch.device("cuda")
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for sph in (False, True):
  for blocks in (dict(tRc=t), dict(tRg=t), dict(tLc=t), dict(tLg=t), dict(tRc=t, tRg=t, tLc=t, tLg=t)):
    kw = dict(sphere=sph, reference_periods=False, **blocks)
    sw, se = SurfWD(**kw), SurfWD(**kw)
    sw.set_warm_start(2); se.set_warm_start(0)
    d0, fl = se.forward(x0)
    sw.set_obsdata(d0 * 1.01); se.set_obsdata(d0 * 1.01)
    xs = np.tile(x0, (nchain, 1)) * (1 + 0.02 * rng.standard_normal((nchain, 2 * n)))
    xs[:, :n] = np.sort(xs[:, :n], axis=1); xs[:, -1] = 0.0
    x = tt(xs); p = tt(0.5 * rng.standard_normal(xs.shape))
    lo, hi = tt(0.7 * xs.min(0)), tt(1.3 * xs.max(0) + 1e-9)
    ctx = None
    for s in range(6):
        mw, gw, dw, fw = sw.misfit_and_grad_device(x)
        me, ge, de, fe = se.misfit_and_grad_device(x)
        ctx = sw._ensure(n)
        r = ((dw - de).abs() / de.abs()).max().item()
        print(sph, list(blocks), "step", s, "maxrel dsyn", f"{r:.2e}", "items", ctx.stat("swd_warm_items"), "declined", ctx.stat("swd_warm_declined_chains"),
              "|dx|max", (dw*0).sum().item())
        x, p = _leapfrog_move(x, p, gw, 0.003, lo, hi)
