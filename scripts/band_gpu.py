"""Band limit of the RF adjoint: gradient with the limit on (default) vs off on the bench's chains, and the step times.
    python scripts/band_gpu.py [config]"""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from rfsurfhmc_amd._lib import K_NAMES
from rfsurfhmc_amd.model.model_rf import ReceiverFunc
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
cfgi = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = bench.CONFIGS[cfgi]; n, nt = cfg["n"], cfg["nt"]; nchain = 8192
dev = torch.device("cuda", 0); t = np.linspace(5, 44, bench.NPER)
j = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(bench.RAY_P, nt, cfg["dt"], bench.GAUSS, bench.TSHIFT, bench.WATER, "P", "freq"), SurfWD(tRc=t))
x_true = bench.true_model(n); drf, dswd, flag = j.forward(x_true); j.set_obsdata(drf, dswd)
ctx = j._ensure(n)
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
xs = bench.make_models(nchain, 991206, n); x = tt(xs)
res = {}
ctx.set_option("rf_band_floor_digits", 0)
for dig in (13, 0, 10, 16, 113):
    ctx.set_option("rf_band_floor_digits", 8 if dig == 113 else 0)
    ctx.set_option("rf_band_limit_digits", dig % 100)
    res[dig] = [o.clone() for o in j.misfit_and_grad_device(x)]
    torch.cuda.synchronize()
g0 = res[0][1]
for dig in (13, 10, 16, 113):
    g = res[dig][1]
    print(f"digits {dig}: grad rel diff vs unlimited max {((g - g0).abs().amax(dim=1) / g0.abs().amax(dim=1)).max().item():.3e}; "
          f"misfit equal {torch.equal(res[dig][0], res[0][0])}, dsyn equal {torch.equal(res[dig][2], res[0][2])}")
bounds = bench.bounds_of(x_true)
for dig, ser in ((113, 0), (113, 1), (13, 0)):
    ctx.set_option("rf_band_floor_digits", 8 if dig > 100 else 0)
    ctx.set_option("rf_band_limit_digits", dig % 100); ctx.set_option("swd_warm_serial", ser)
    st = j.flow_state(tt(xs), torch.full((nchain,), 0.002, dtype=torch.float64, device=dev), tt(bounds))
    st["p"].copy_(tt(0.5 * np.random.default_rng(7).standard_normal(xs.shape))); st["rem"].fill_(1 << 30); st["fresh"].fill_(1)
    for _ in range(40): j.flow_step(st)
    ctx.check(ctx.L.rfs_synchronize(ctx.h)); torch.cuda.synchronize()
    ctx.check(ctx.L.rfs_enable_timing(ctx.h, 1))
    for _ in range(5): j.flow_step(st)
    ms = np.zeros(len(K_NAMES)); cnt = np.zeros(len(K_NAMES), dtype=np.int32)
    ctx.check(ctx.L.rfs_kernel_ms_sum(ctx.h, ms.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p)))
    ctx.check(ctx.L.rfs_enable_timing(ctx.h, 0))
    torch.cuda.synchronize(); t0 = time.perf_counter(); K = 30
    for _ in range(K): j.flow_step(st)
    ctx.check(ctx.L.rfs_synchronize(ctx.h)); torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / K * 1e3
    print(f"config {cfgi} band digits {dig} serial {ser}: {el:.3f} ms/step = {nchain / el * 1e3:.0f} evals/s", {k: round(ms[i] / max(cnt[i], 1), 3) for i, k in enumerate(K_NAMES)})
