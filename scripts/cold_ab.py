"""The search without a prediction ("swd_cold_scan") against the sequential search for the chains the warm start declines:
configs[0]'s plugin (SWD only, 10 layers, 36 Rc + 36 Rg), chains started from the sampler's own random models at dt 0.1 --
the same accepted end points and misfits?  hand-backs, ms per device step.
    python3 scripts/cold_ab.py [steps=300] [nchains=1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
thk = np.array([3., 3, 4, 5, 5, 6, 7, 8, 10, 0]); vs = np.linspace(2.9, 4.6, 10)
t = np.arange(5., 41.)
x0 = np.hstack((vs, thk))
m = SurfWD(tRc=t, tRg=t, device=0)
d, flag = m.forward(x0); assert flag
m.set_obsdata(d)
bounds = bench.bounds_of(x0)
ctx = m._ensure(10)
names = ("swd_warm_declined_chains", "swd_cold_chains", "swd_cold_secular_evals", "swd_exact_declined_chains", "swd_warm_secular_evals") + \
        tuple(f"swd_cold_fail_{i}" for i in range(34, 40)) + tuple(f"swd_warm_cause_{i}" for i in range(4, 12)) + tuple(f"swd_exact_cause_{i}" for i in range(1, 8))
out = {}
FIRST = int(os.environ.get("COLD_FIRST", "0"))
for cold in (0, -1, 0, -1):
    ctx.set_option("swd_cold_scan", cold); ctx.set_option("swd_cold_first", FIRST if cold else 0)
    c0 = [ctx.stat(k) for k in names]
    smp = HamitonianMC(m, bounds, 0.1, [5, 20], 10, 991206, 800, 0, myrank=0, name="c0", outdir=None, nchains=nch, verbose=False, store_syn=False)
    marks = {}
    def hook(s, st):
        if s == 40: torch.cuda.synchronize(); marks["t0"] = time.perf_counter(); marks["e0"] = ctx.stat("flow_chain_steps")
        if s == 40 + K: ctx.check(ctx.L.rfs_synchronize(ctx.h)); torch.cuda.synchronize(); marks["t1"] = time.perf_counter(); marks["e1"] = ctx.stat("flow_chain_steps")
    mis = smp.sample_flow(max_steps=40 + K + 1, step_hook=hook)
    el, ev = marks["t1"] - marks["t0"], marks["e1"] - marks["e0"]
    st = [ctx.stat(k) - v for k, v in zip(names, c0)]
    print(f"cold {cold:2d}: {el / K * 1e3:.3f} ms per device step, {el / max(ev, 1) * 1e3:.3f} ms per evaluation; " +
          ", ".join(f"{k[4:]} {v}" for k, v in zip(names, st)) + f"; accepted {[int(a.sum()) for a in smp.live_counts]}")
    out.setdefault(cold, (np.atleast_2d(np.asarray(mis)), np.asarray(smp.x_cache), np.asarray(smp.naccepted), np.asarray(smp.ntrajectories)))
a, b = out[0], out[-1]
# (no burn-in: every accepted end point and its misfit is stored; how many trajectories fit into the run depends on the host's timing,
# their results do not -- the accepted end points the two runs have in common, chain by chain, one by one)
na, nb = a[2], b[2]
same = 0
for c in range(nch):
    k = int(min(na[c], nb[c]))
    diff = (a[0][c, :k] != b[0][c, :k]) | (a[1][c, :k] != b[1][c, :k]).any(1)
    if not diff.any():
        same += 1
    else:
        j = int(np.argmax(diff))
        print(f"chain {c}: first differing accepted end point {j} of {k}, misfit rel. diff {abs(a[0][c, j] - b[0][c, j]) / abs(a[0][c, j]):.3e}")
print(f"chains identical throughout: {same} of {nch}; accepted {na.tolist()} / {nb.tolist()}")
