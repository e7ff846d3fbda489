"""Small fixed workload for rocprofv3: leapfrog (flow) steps of 8192 chains at one of bench.py's configurations.
    python3 scripts/prof_run.py [config=1] [steps=10] [warm_start=1]
The first call (start models) goes through the reference-semantics root search, the `steps` calls after it through the
warm-started one (warm_start = 1) -- scripts/pmc_summary.py divides the counters by the calls each kernel ran in."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rfsurfhmc_amd.model.model_rf import ReceiverFunc
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
cfg = bench.CONFIGS[int(sys.argv[1]) if len(sys.argv) > 1 else 1]
nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 10
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n, nt, nchain = cfg["n"], cfg["nt"], 8192
t = np.linspace(5, 44, bench.NPER)
joint = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(bench.RAY_P, nt, cfg["dt"], bench.GAUSS, bench.TSHIFT, bench.WATER, "P", "freq"), SurfWD(tRc=t))
joint.set_warm_start(warm)
import os
SERIAL = os.environ.get("RFS_SERIAL") == "1"      # one stream: every kernel alone on the chip (clean per-kernel durations)
x_true = bench.true_model(n)
drf, dswd, flag = joint.forward(x_true); joint.set_obsdata(drf, dswd)
dev = torch.device("cuda")
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bounds = bench.bounds_of(x_true)
xs = np.clip(bench.make_models(nchain, 991206, n), bounds[:, 0], bounds[:, 1])
st = joint.flow_state(tt(xs), torch.full((nchain,), 0.002, dtype=torch.float64, device=dev), tt(bounds))
st["p"].copy_(tt(0.5 * np.random.default_rng(7).standard_normal(xs.shape))); st["rem"].fill_(1 << 30); st["fresh"].fill_(1)
if SERIAL:
    joint._ensure(n).set_option("swd_warm_serial", 1)
for _ in range(nrep + 1):
    joint.flow_step(st)
torch.cuda.synchronize()
ctx = joint._ensure(n); ctx.L.rfs_synchronize(ctx.h)
print("done", nrep + 1, "calls;", "declined", ctx.stat("swd_warm_declined_chains"))
