"""Debug: k_swd_exact_coop's evaluations against the single lane's (library built with -DRFS_DEBUG_COOP, RFSURF_LIB=...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
cfg = bench.CONFIGS[1]
joint, x_true, bounds = bench.make_joint(cfg, 0)
ctx = joint._ensure(30)
ctx.set_option("swd_exact_coop", 2)
nc = 256
s = HamitonianMC(joint, bounds, 0.05, [5, 20], 2, 991206, 30, 4, myrank=0, name="t", outdir=None, nchains=nc, verbose=False)
s.sample_flow(x_init=bench.make_models(nc, 4, 30), max_steps=60, async_handback=False)
print("compared", ctx.stat("wstat_16"), "mismatches", ctx.stat("wstat_3"), "exact evals", ctx.stat("swd_exact_secular_evals"),
      "exact declined", ctx.stat("swd_exact_declined_chains"), {k: ctx.stat(f"swd_exact_cause_{k}") for k in range(1, 8)})
