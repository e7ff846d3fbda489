"""How many chains per device step go through the host path (done = 1) / the device restart (done >= 2) in the bench's headline run?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rfsurfhmc_amd.pyhmc import hmc as H
cnt = {"host": [], "why": {}}
orig = H.HamitonianMC.__dict__.get("sample_flow")
import rfsurfhmc_amd.pyhmc._batched as B
orig_run = B._run_flow
def patched(model, st, process_done, active, fetch_syn, pipeline, max_steps, step_hook, restart):
    def pd(idx, res):
        cnt["host"].append(len(idx))
        ok = res["ok"]
        cnt["why"]["ok0"] = cnt["why"].get("ok0", 0) + int((ok == 0).sum())
        cnt["why"]["ok1"] = cnt["why"].get("ok1", 0) + int((ok != 0).sum())
        return process_done(idx, res)
    return orig_run(model, st, pd, active, fetch_syn, pipeline, max_steps, step_hook, restart)
B._run_flow = patched
cfg = bench.CONFIGS[1]
dev = torch.device("cuda:0")
joint, x_true, bounds = bench.make_joint(cfg, 0)
rep, *_ = bench.sampler_leg(cfg, 1, joint, x_true, bounds, 8192, 0, dev, 150, 250, lambda: torch.cuda.synchronize(), kind="hmc", dt=bench.TUNED_DT, mode="reference_roots", groups=False)
h = np.array(cnt["host"])
print("ms/step", rep["ms_per_step"], "host-path calls", len(h), "of ~400 steps; chains per call: mean", h.mean() if len(h) else 0, "max", h.max() if len(h) else 0, cnt["why"])
