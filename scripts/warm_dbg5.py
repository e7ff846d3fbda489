import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
cfg = bench.CONFIGS[1]; n = cfg["n"]; nchain = 8192
joint, x_true, bounds = bench.make_joint(cfg, 0)
ctx = joint._ensure(n); dev = torch.device("cuda")
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
xs = bench.make_models(nchain, 991206, n)
st = joint.flow_state(tt(xs), torch.full((nchain,), 0.002, dtype=torch.float64, device=dev), tt(bounds))
st["p"].copy_(tt(0.5 * np.random.default_rng(7).standard_normal(xs.shape))); st["rem"].fill_(1 << 30); st["fresh"].fill_(1)
prev = {k: 0 for k in range(4, 12)}
for s in range(80):
    joint.flow_step(st)
    cur = {k: ctx.stat(f"swd_warm_cause_{k}") for k in range(4, 12)}
    d = {k: cur[k] - prev[k] for k in cur if cur[k] != prev[k]}
    if d and s > 2: print(s, d)
    prev = cur
