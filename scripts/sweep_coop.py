"""Sweep of the cooperative root-search block shape (rfs_set_option swd_coop_shape / swd_coop_blocks_per_cu) at the
bench workload: ms per evaluation of all chains, ms of the root-search group, results compared bit for bit."""
import sys, time, ctypes; sys.path.insert(0, '.')
import numpy as np, torch
import bench
from rfsurfhmc_amd._lib import K_NAMES
from rfsurfhmc_amd.model.model_rf import ReceiverFunc
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
cfgid = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nchain = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
cfg = bench.CONFIGS[cfgid]
n, nt = cfg["n"], cfg["nt"]
t = np.linspace(5, 44, bench.NPER)
joint = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(bench.RAY_P, nt, cfg["dt"], bench.GAUSS, bench.TSHIFT, bench.WATER, "P", "freq"), SurfWD(tRc=t))
drf, dswd, flag = joint.forward(bench.true_model(n)); joint.set_obsdata(drf, dswd)
x = torch.from_numpy(bench.make_models(nchain, 991206, n=n)).cuda()
ctx = joint._ensure(n)
ref = None
combos = [(81, 1), (42, 2), (42, 1), (82, 1), (44, 2), (44, 3), (44, 0)] if len(sys.argv) <= 3 else [tuple(int(v) for v in a.split(',')) for a in sys.argv[3:]]
for shape, pcu in combos:
    ctx.check(ctx.L.rfs_set_option(ctx.h, b"swd_coop_shape", shape))
    ctx.check(ctx.L.rfs_set_option(ctx.h, b"swd_coop_blocks_per_cu", pcu))
    for split in (1, 0):
        ctx.check(ctx.L.rfs_set_option(ctx.h, b"cu_split", split))
        for _ in range(2): out = joint.misfit_and_grad_device(x)
        torch.cuda.synchronize(); ctx.L.rfs_synchronize(ctx.h)
        ctx.L.rfs_enable_timing(ctx.h, 1)
        t0 = time.perf_counter()
        for _ in range(8): out = joint.misfit_and_grad_device(x)
        ctx.L.rfs_synchronize(ctx.h); torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / 8
        ms = np.zeros(len(K_NAMES)); cnt = np.zeros(len(K_NAMES), dtype=np.int32)
        ctx.L.rfs_kernel_ms_sum(ctx.h, ms.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p))
        ctx.L.rfs_enable_timing(ctx.h, 0)
        res = [o.cpu().numpy() for o in out]
        if ref is None: ref = res
        same = all(np.array_equal(a, b) for a, b in zip(res, ref))
        per = {k: round(ms[i] / max(cnt[i], 1), 3) for i, k in enumerate(K_NAMES)}
        print(f"shape {shape} per_cu {pcu} cu_split {split}: {el*1e3:7.3f} ms/eval  {nchain/el:9.0f} evals/s  identical {same}  {per}", flush=True)
