"""Why chains are handed back to the full search in the bench's headline run: swd_warm_cause_4..11 over the timed window."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
cfg = bench.CONFIGS[1]
dev = torch.device("cuda:0")
joint, x_true, bounds = bench.make_joint(cfg, 0)
ctx = joint._ensure(cfg["n"])
names = [f"swd_warm_cause_{i}" for i in range(4, 12)] + [f"swd_exact_cause_{i}" for i in range(1, 8)] + ["swd_warm_fail_no_change", "swd_warm_fail_other", "swd_warm_wide_chains", "swd_exact_secular_evals", "swd_warm_items", "swd_warm_secular_evals"] + ["swd_warm_declined_chains", "swd_exact_declined_chains", "swd_warm_walked_chains", "flow_chain_steps", "swd_warm_search_evals", "swd_warm_search_evals_slowest_lane", "swd_warm_search_lanes", "swd_warm_passed_on_1", "swd_warm_passed_on_2", "swd_warm_passed_on_3"]
snap = {}
orig = bench.sampler_leg
import time
K, burn = 150, 250
import rfsurfhmc_amd.pyhmc._batched as B
orig_run = B._run_flow
def patched(model, st, process_done, active, fetch_syn, pipeline, max_steps, step_hook, restart):
    def hook(s, st_):
        if s == burn: snap[0] = {k: ctx.stat(k) for k in names}
        if s == burn + K: snap[1] = {k: ctx.stat(k) for k in names}
        if step_hook: step_hook(s, st_)
    return orig_run(model, st, process_done, active, fetch_syn, pipeline, max_steps, hook, restart)
B._run_flow = patched
rep, *_ = bench.sampler_leg(cfg, 1, joint, x_true, bounds, 8192, 0, dev, K, burn, lambda: torch.cuda.synchronize(), kind="hmc", dt=bench.TUNED_DT, mode="reference_roots", groups=False)
print("ms/step", rep["ms_per_step"])
print({k: (snap[1][k] - snap[0][k]) / K for k in names})
d = {k: (snap[1][k] - snap[0][k]) / K for k in names}
if d["swd_warm_search_lanes"]:
    print("warm search: %.2f evaluations per lane, %.2f executed per lane (the wavefront's slowest lane x 64)" % (d["swd_warm_search_evals"] / d["swd_warm_search_lanes"], 64 * d["swd_warm_search_evals_slowest_lane"] / d["swd_warm_search_lanes"]))
print("causes: 4 no valid previous evaluation / forced, 5 refused (move too large), 6 warm search failed, 7 root above the fastest layer; 8 / 10 degenerate start point (later / first period), 9 / 11 branch test (later / first period)")
