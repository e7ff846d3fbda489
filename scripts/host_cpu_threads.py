"""Where a rank's host CPU time goes: per-thread CPU seconds (utime + stime from /proc/self/task) over 200 device steps of the
bench's sampler run.   python3 scripts/host_cpu_threads.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC

def threads():
    out = {}
    tck = os.sysconf("SC_CLK_TCK")
    for t in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{t}/stat").read()
            name = f[f.index("(") + 1:f.rindex(")")]
            r = f[f.rindex(")") + 2:].split()
            out[int(t)] = (name, int(r[11]) / tck, int(r[12]) / tck)
        except Exception:
            pass
    return out

cfg = bench.CONFIGS[1]
joint, x_true, bounds = bench.make_joint(cfg, 0)
smp = HamitonianMC(joint, bounds, bench.TUNED_DT, [5, 20], 10, 991206, 200, 20, myrank=0, name="t", outdir=None, nchains=8192, verbose=False, store_syn=False)
m = {}
def hook(s, st):
    if s == 300: m["a"] = threads(); m["t0"] = time.perf_counter()
    if s == 500: m["b"] = threads(); m["t1"] = time.perf_counter()
smp.sample_flow(x_init=bench.make_models(8192, 991206, 30), max_steps=502, step_hook=hook)
el = m["t1"] - m["t0"]
print(f"200 steps in {el:.3f} s; main thread {os.getpid()}")
for t, (name, u, s) in sorted(m["b"].items()):
    u0, s0 = m["a"].get(t, (name, 0, 0))[1:]
    if (u - u0) + (s - s0) > 0.01:
        print(f"  thread {t} {name:16s} user {u - u0:.2f} s  sys {s - s0:.2f} s  = {(u - u0 + s - s0) / el:.2f} cores")
