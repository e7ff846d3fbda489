"""The search without a prediction against the sequential search over many sampler seeds: configs[0]'s plugin, chains from the
sampler's own random start models at dt 0.1 -- same samples, misfits and accept counts?
    python3 scripts/cold_seeds.py [nseeds=12] [steps=200]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
# (device_restart / pipeline off: a chain that finishes a trajectory waits exactly one device step for the host, whatever the host's
# timing -- the number of trajectories inside a fixed number of device steps is then the same in every run)
SEEDS = [int(v) for v in os.environ.get("COLD_SEEDS", "").split(",") if v] or list(range(NS))
NCH = [int(v) for v in os.environ.get("COLD_NCH", "1,8,24").split(",")]
thk = np.array([3., 3, 4, 5, 5, 6, 7, 8, 10, 0]); vs = np.linspace(2.9, 4.6, 10)
t = np.arange(5., 41.)
x0 = np.hstack((vs, thk))
m = SurfWD(tRc=t, tRg=t, device=0)
d, flag = m.forward(x0); assert flag
m.set_obsdata(d)
bounds = bench.bounds_of(x0)
ctx = m._ensure(10)
names = ("swd_warm_declined_chains", "swd_cold_chains")
tot = dict(runs=0, same=0, seq_off=0, seq_on=0, cold=0)
for seed in SEEDS:
    for nch in NCH:
        out = {}
        for cold in (0, -1):
            ctx.set_option("swd_cold_scan", cold)
            c0 = [ctx.stat(k) for k in names]
            smp = HamitonianMC(m, bounds, 0.1, [5, 20], 10, 1000 + 17 * seed, 800, 200, myrank=0, name="c0", outdir=None, nchains=nch, verbose=False, store_syn=False)
            mis = smp.sample_flow(max_steps=K, pipeline=False, device_restart=False)
            out[cold] = (np.asarray(mis), np.asarray(smp.x_cache), np.asarray(smp.naccepted), [ctx.stat(k) - v for k, v in zip(names, c0)])
        same = all(np.array_equal(out[0][i], out[-1][i]) for i in range(3))
        tot["runs"] += 1; tot["same"] += int(same); tot["seq_off"] += out[0][3][0]; tot["seq_on"] += out[-1][3][0]; tot["cold"] += out[-1][3][1]
        if not same:
            dm = np.abs(out[0][0] - out[-1][0]) / (np.abs(out[0][0]) + 1e-300) if out[0][0].shape == out[-1][0].shape else None
            print(f"seed {seed} nchains {nch}: DIFFER  (misfit rel. max {None if dm is None else float(np.nanmax(dm)):.3e}, accepted {out[0][2].tolist()} / {out[-1][2].tolist()})")
print(f"{tot['runs']} sampler runs of {K} device steps (1, 8 and 24 chains, {NS} seeds): identical samples, misfits and accept counts in {tot['same']}; "
      f"chain evaluations through the sequential search {tot['seq_off']} -> {tot['seq_on']}, through the search without a prediction {tot['cold']}")
