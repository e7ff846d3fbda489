"""The search without a prediction against the sequential search over many sampler seeds: configs[0]'s plugin, chains from the
sampler's own random start models at dt 0.1 -- the same accepted end points and misfits, one by one?
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
            smp = HamitonianMC(m, bounds, 0.1, [5, 20], 10, 1000 + 17 * seed, 800, 0, myrank=0, name="c0", outdir=None, nchains=nch, verbose=False, store_syn=False)
            mis = smp.sample_flow(max_steps=K)
            out[cold] = (np.atleast_2d(np.asarray(mis)), np.asarray(smp.x_cache), np.asarray(smp.naccepted), [ctx.stat(k) - v for k, v in zip(names, c0)])
        # (no burn-in: every accepted end point and its misfit is stored in the order the chain accepted them; how many trajectories fit
        # into K device steps depends on the host's timing, their results do not: compare what the two runs have in common)
        na, nb = out[0][2], out[-1][2]
        bad = []
        for c in range(nch):
            k = int(min(na[c], nb[c]))
            if not (np.array_equal(out[0][0][c, :k], out[-1][0][c, :k]) and np.array_equal(out[0][1][c, :k], out[-1][1][c, :k])):
                j = int(np.argmax((out[0][0][c, :k] != out[-1][0][c, :k]) | (out[0][1][c, :k] != out[-1][1][c, :k]).any(1)))
                bad.append((c, j, float(abs(out[0][0][c, j] - out[-1][0][c, j]) / abs(out[0][0][c, j]))))
        same = not bad
        tot["runs"] += 1; tot["same"] += int(same); tot["seq_off"] += out[0][3][0]; tot["seq_on"] += out[-1][3][0]; tot["cold"] += out[-1][3][1]
        tot["acc"] = tot.get("acc", 0) + int(np.minimum(na, nb).sum())
        if not same:
            print(f"seed {seed} nchains {nch}: first differing accepted end point (chain, index, misfit rel. diff) {bad}")
print(f"{tot['runs']} sampler runs of {K} device steps (1, 8 and 24 chains, {NS} seeds): every accepted end point and misfit identical in {tot['same']} ({tot['acc']} accepted trajectories compared); "
      f"chain evaluations through the sequential search {tot['seq_off']} -> {tot['seq_on']}, through the search without a prediction {tot['cold']}")
