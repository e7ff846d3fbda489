"""Small fixed workloads for rocprofv3 at one of bench.py's configurations, 8192 chains.
    python3 scripts/prof_run.py <config=1> <steps=60> [hmc|flow] [warm_start=1]
hmc (default): the bench's headline workload -- a sampler run (HamitonianMC.sample_flow at the configuration's step size, or
    HMCDualAveraging.sample_flow for configs[3]) of <steps> device steps that CONTINUES burned-in chains: the first call for a
    configuration burns them in (300 steps, unprofiled use: run it once without the profiler) and keeps the models in
    gpurun_out/burned_c<config>.npy, later calls load them.  scripts/pmc_summary.py divides the counters by the device steps
    (k_prep_joint dispatches).
flow: rounds 1-3's workload -- rfs_flow_step on never-ending trajectories of the random start models at dt = 0.002."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
cfgi = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = bench.CONFIGS[cfgi]
nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 60
mode = sys.argv[3] if len(sys.argv) > 3 else "hmc"
warm = int(sys.argv[4]) if len(sys.argv) > 4 else 1
n, nt, nchain = cfg["n"], cfg["nt"], 8192
joint, x_true, bounds = bench.make_joint(cfg, 0)
joint.set_warm_start(warm)
ctx = joint._ensure(n)
SERIAL = os.environ.get("RFS_SERIAL") == "1"      # one stream: every kernel alone on the chip (clean per-kernel durations)
if SERIAL:
    ctx.set_option("swd_warm_serial", 1)
dev = torch.device("cuda")
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
if mode == "flow":
    xs = np.clip(bench.make_models(nchain, 991206, n), bounds[:, 0], bounds[:, 1])
    st = joint.flow_state(tt(xs), torch.full((nchain,), 0.002, dtype=torch.float64, device=dev), tt(bounds))
    st["p"].copy_(tt(0.5 * np.random.default_rng(7).standard_normal(xs.shape))); st["rem"].fill_(1 << 30); st["fresh"].fill_(1)
    for _ in range(nrep + 1):
        joint.flow_step(st)
else:
    from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
    from rfsurfhmc_amd.pyhmc.hmcda import HMCDualAveraging
    da = cfg["sampler"] == "da"
    dt = cfg.get("hmc_dt", bench.TUNED_DT)

    def sampler(nsamp):
        if da:
            return HMCDualAveraging(joint, bounds, 0.1, 10, max(10, nsamp // 10 + 1), 0.65, 991206, nsamp, 20, myrank=0, name="p",
                                    outdir=None, nchains=nchain, verbose=False, store_syn=False)
        return HamitonianMC(joint, bounds, dt, [5, 20], 10, 991206, nsamp, 20, myrank=0, name="p", outdir=None, nchains=nchain,
                            verbose=False, store_syn=False)
    path = os.path.join(ROOT, "gpurun_out", f"burned_c{cfgi}.npy")
    if os.path.exists(path):
        xb = np.load(path)
    else:
        if da:
            rs = np.random.default_rng(3)
            xs = np.clip(x_true[None, :] * (1 + 0.02 * rs.standard_normal((nchain, 2 * n))), bounds[:, 0], bounds[:, 1])
            xs[:, :n] = np.sort(xs[:, :n], axis=1)
        else:
            xs = bench.make_models(nchain, 991206, n)
        keep = {}
        sampler(120).sample_flow(x_init=xs, max_steps=301, step_hook=lambda s, st: keep.__setitem__("x", st["x"].clone()) if s == 300 else None)
        xb = keep["x"].cpu().numpy()
        os.makedirs(os.path.dirname(path), exist_ok=True)
        np.save(path, xb)
        print("burned in", path)
    sampler(nrep // 4 + 20).sample_flow(x_init=xb, max_steps=nrep)
torch.cuda.synchronize()
ctx.L.rfs_synchronize(ctx.h)
print("done", nrep, "device steps;", "handed back", ctx.stat("swd_warm_declined_chains"))
