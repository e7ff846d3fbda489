"""Wild models one after the other (nothing to continue) through the plugin with warm start 2: roots against the oracle, by path.
    python3 scripts/cold_wild.py [nbatches=12]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from oracle import oracle as orc
from rfsurfhmc_amd.model.model_surf import SurfWD
orc.build(ref=False)
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 12
thk = np.array([3., 3, 4, 5, 5, 6, 7, 8, 10, 0]); vs = np.linspace(2.9, 4.6, 10)
tt = np.arange(5., 41.)
x0 = np.hstack((vs, thk))
m = SurfWD(tRc=tt, tRg=tt, device=0)
d0, flag = m.forward(x0); m.set_obsdata(d0)
b0 = bench.bounds_of(x0)
ctx = m._ensure(10)
o = orc.SurfWD(tRc=tt, tRg=tt); o.set_obsdata(d0)
dev = torch.device("cuda")
m.set_warm_start(2)
names = ("swd_warm_declined_chains", "swd_cold_chains", "swd_exact_declined_chains", "swd_warm_wide_chains")
ref = {}
for tag, opts in (("cold off", {"swd_cold_scan": 0}), ("cold on", {"swd_cold_scan": -1, "swd_cold_first": 0}), ("cold first", {"swd_cold_scan": -1, "swd_cold_first": 8}),
                  ("cold first, runup 4", {"swd_cold_scan": -1, "swd_cold_first": 8, "swd_exact_runup": 4})):
    for k, v in opts.items(): ctx.set_option(k, v)
    ctx.set_option("swd_warm_reset", 1)
    rng = np.random.default_rng(11)
    c0 = [ctx.stat(k) for k in names]
    nroot = nsame = nbad = 0; worst = 0.0; offchains = 0; nch = 0
    for it in range(NB):
        xs = b0[:, 0] + (b0[:, 1] - b0[:, 0]) * rng.random((8, 20)); xs[:, 19] = 0.0
        mis, g, d, f = m.misfit_and_grad_device(torch.from_numpy(xs).to(dev))
        d, f = d.cpu().numpy(), f.cpu().numpy() != 0
        for i in range(8):
            key = (it, i)
            if key not in ref: ref[key] = o.misfit_and_grad(xs[i])
            mo, go, do, fo = ref[key]
            if bool(fo) != bool(f[i]): nbad += 1; continue
            if not fo or it == 0: continue
            rel = np.abs(d[i, :36] - do[:36]) / do[:36]
            nroot += 36; nsame += int((d[i, :36] == do[:36]).sum()); worst = max(worst, float(rel.max())); nch += 1
            offchains += int((d[i, :36] != do[:36]).any())
    st = [ctx.stat(k) - v for k, v in zip(names, c0)]
    print(f"{tag}: {nsame}/{nroot} bit-identical ({nroot - nsame} off in {offchains} of {nch} chain evaluations), worst {worst:.2e}, flag mismatches {nbad}; " + ", ".join(f"{k[4:]} {v}" for k, v in zip(names, st)))
    ctx.set_option("swd_exact_runup", 2)
