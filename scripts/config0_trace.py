"""configs[0] (one chain, as scripts/config0_flow.py): the model, the roots and the hand-back counter after every device step ->
gpurun_out/config0_trace.npz (to look at on the CPU with the oracle: why does the warm search decline a step?)
    python3 scripts/config0_trace.py [steps=340]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
K = int(sys.argv[1]) if len(sys.argv) > 1 else 340
thk = np.array([3., 3, 4, 5, 5, 6, 7, 8, 10, 0]); vs = np.linspace(2.9, 4.6, 10)
t = np.arange(5., 41.)
x0 = np.hstack((vs, thk))
m = SurfWD(tRc=t, tRg=t, device=0)
d, flag = m.forward(x0); assert flag
m.set_obsdata(d)
smp = HamitonianMC(m, bench.bounds_of(x0), 0.1, [5, 20], 10, 991206, 800, 200, myrank=0, name="c0", outdir=None, nchains=1, verbose=False, store_syn=False)
ctx = m._ensure(10)
xs, roots, declined, causes = [], [], [], []
def hook(s, st):
    ctx.check(ctx.L.rfs_synchronize(ctx.h)); torch.cuda.synchronize()
    xs.append(st["x"].detach().cpu().numpy().copy())
    roots.append(ctx.last_roots(1).copy() if s > 0 else np.zeros((1, 1)))
    declined.append(ctx.stat("swd_warm_declined_chains"))
    causes.append([ctx.stat(f"swd_warm_cause_{k}") for k in range(4, 12)] + [ctx.stat("swd_warm_fail_no_change"), ctx.stat("swd_warm_fail_other")])
smp.sample_flow(max_steps=K, step_hook=hook)
os.makedirs("gpurun_out", exist_ok=True)
nr = max(r.shape[1] for r in roots)
R = np.zeros((len(roots), nr)); 
for i, r in enumerate(roots): R[i, :r.shape[1]] = r[0]
np.savez("gpurun_out/config0_trace.npz", x=np.array(xs)[:, 0], roots=R, declined=np.array(declined), causes=np.array(causes), dobs=d, t=t)
print("steps", len(xs), "declined", declined[-1])
