import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from test_gpu_warm import _bench_joint, _leapfrog_move
n, nt, nchain = 30, 512, 1024
dev = torch.device("cuda"); tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rng = np.random.default_rng(11)
xs = bench.make_models(nchain, 5, n)
wild = rng.random(nchain) < 0.3
for i in np.nonzero(wild)[0]:
    xs[i, :n] = rng.permutation(xs[i, :n])
bounds = np.stack([np.r_[np.full(n, 1.5), np.full(n, 0.0)], np.r_[np.full(n, 5.0), np.full(n, 3.0)]], axis=1)
lo, hi = tt(bounds[:, 0]), tt(bounds[:, 1])
jw, _ = _bench_joint(2); je, _ = _bench_joint(0)
x = tt(xs); p = tt(0.5 * rng.standard_normal(xs.shape))
out = {}
xprev = None
for s in range(8):
    mw, gw, dw, fw = jw.misfit_and_grad_device(x)
    me, ge, de, fe = je.misfit_and_grad_device(x)
    ok = (fe != 0) & (fw != 0)
    r = ((dw[:, nt:] - de[:, nt:]).abs() / de[:, nt:].abs().clamp_min(1e-30))
    r[~ok] = 0
    if r.max().item() > 1.2e-6:
        ch = int(r.amax(dim=1).argmax()); k = int(r[ch].argmax())
        print("step", s, "chain", ch, "period", k, "rel", r[ch, k].item(), "warm", dw[ch, nt + k].item(), "exact", de[ch, nt + k].item())
        out[f"s{s}_xprev"] = xprev[ch].cpu().numpy(); out[f"s{s}_x"] = x[ch].cpu().numpy()
        out[f"s{s}_cw"] = dw[ch, nt:].cpu().numpy(); out[f"s{s}_ce"] = de[ch, nt:].cpu().numpy(); out[f"s{s}_cprev"] = cprev[ch].cpu().numpy()
    xprev = x.clone(); cprev = dw[:, nt:].clone()
    bad = fe == 0
    x, p = _leapfrog_move(x, p, torch.where(bad[:, None], torch.zeros_like(gw), gw), 0.002, lo, hi)
np.savez(os.path.join(ROOT, "gpurun_out", "wild_case.npz"), **out)
