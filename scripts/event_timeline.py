"""Where in a device step each kernel group runs -- WITHOUT a profiler attached (rfs_kernel_timeline: HIP events of the
library's own group timers, offsets against the step's first launch), and how long the caller's stream idles between two
steps (wall time per step - the step's span on that stream).   python3 scripts/event_timeline.py [steps=100] [serial=0]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rfsurfhmc_amd._lib import K_NAMES
from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
cfg = bench.CONFIGS[int(os.environ.get("TL_CONFIG", "1"))]
dev = torch.device("cuda:0")
joint, x_true, bounds = bench.make_joint(cfg, 0)
n = cfg["n"]
ctx = joint._ensure(n)
burn = 300
smp = HamitonianMC(joint, bounds, cfg.get("hmc_dt", bench.TUNED_DT), [5, 20], 10, 991206, 200, 20, myrank=0, name="tl", outdir=None, nchains=8192,
                   verbose=False, store_syn=False)
marks = {}
def hook(s, st):
    if s == burn:
        ctx.check(ctx.L.rfs_synchronize(ctx.h)); torch.cuda.synchronize()
        ctx.check(ctx.L.rfs_enable_timing(ctx.h, 1))
        marks["t0"] = time.perf_counter()
    if s == burn + K:
        ctx.check(ctx.L.rfs_synchronize(ctx.h)); torch.cuda.synchronize()
        marks["t1"] = time.perf_counter()
        a = np.zeros(len(K_NAMES)); b = np.zeros(len(K_NAMES)); c = np.zeros(len(K_NAMES), dtype=np.int32)
        ctx.check(ctx.L.rfs_kernel_timeline(ctx.h, a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p), c.ctypes.data_as(ctypes.c_void_p)))
        marks["tl"] = (a, b, c)
        ctx.check(ctx.L.rfs_enable_timing(ctx.h, 0))
smp.sample_flow(x_init=bench.make_models(8192, 991206, n), max_steps=burn + K + 1, step_hook=hook)
a, b, c = marks["tl"]
wall = (marks["t1"] - marks["t0"]) / K * 1e3
print(f"# {cfg['name']}: {K} device steps of HamitonianMC.sample_flow, all group timers on (event pairs cost ~0.1 ms per step)")
print(f"# wall time per step {wall:.3f} ms")
print(f"# {'group':12s} {'start':>8s} {'end':>8s} {'dur':>8s}   (ms after the step's first launch, mean over {int(c[-1])} steps)")
rows = sorted([(a[i] / max(c[i], 1), b[i] / max(c[i], 1), k, c[i]) for i, k in enumerate(K_NAMES) if c[i] > 0])
for s0, s1, k, cc in rows:
    print(f"  {k:12s} {s0:8.3f} {s1:8.3f} {s1 - s0:8.3f}   x{cc / max(c[-1], 1):.2f} per step")
span = b[-1] / max(c[-1], 1)
print(f"# step span on the caller's stream {span:.3f} ms -> {wall - span:.3f} ms per step between the end of one step and the first launch of the next")
