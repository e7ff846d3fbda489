"""configs[0] as the bench's config0 leg runs it (ONE chain, SWD-only plugin, 36 Rc + 36 Rg, HamitonianMC.sample_flow at dt 0.1):
ms per device step without a profiler, or -- under `rocprofv3 --kernel-trace` -- the launches of one step.
    python3 scripts/config0_flow.py [steps=300] [nchains=1]"""
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
smp = HamitonianMC(m, bounds, 0.1, [5, 20], 10, 991206, 800, 200, myrank=0, name="c0", outdir=None, nchains=nch, verbose=False, store_syn=False)
ctx = m._ensure(10)
for kv in filter(None, os.environ.get("RFS_OPTS", "").split(",")):
    k_, v_ = kv.split("="); ctx.set_option(k_, int(v_))
marks = {}
burn = 40
def hook(s, st):
    if s == burn:
        torch.cuda.synchronize(); marks["e0"] = ctx.stat("flow_chain_steps"); marks["t0"] = time.perf_counter()
    if s == burn + K:
        ctx.check(ctx.L.rfs_synchronize(ctx.h)); torch.cuda.synchronize()
        marks["t1"] = time.perf_counter(); marks["e1"] = ctx.stat("flow_chain_steps")
smp.sample_flow(max_steps=burn + K + 1, step_hook=hook)
el, ev = marks["t1"] - marks["t0"], marks["e1"] - marks["e0"]
print(f"{nch} chain(s): {el / K * 1e3:.3f} ms per device step, {ev} chain steps, {el / max(ev, 1) * 1e3:.3f} ms per evaluation; "
      f"handed back {ctx.stat('swd_warm_declined_chains')}, exact declined {ctx.stat('swd_exact_declined_chains')}")
print("warm causes", {k: ctx.stat(f"swd_warm_cause_{k}") for k in range(4, 12)}, "exact causes", {k: ctx.stat(f"swd_exact_cause_{k}") for k in range(1, 8)},
      "fail_no_change", ctx.stat("swd_warm_fail_no_change"), "fail_other", ctx.stat("swd_warm_fail_other"), "walked", ctx.stat("swd_warm_walked_chains"),
      "wide", ctx.stat("swd_warm_wide_chains"))
print("accepted / trajectories", [int(a.sum()) for a in smp.live_counts])
