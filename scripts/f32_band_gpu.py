"""The column sweep's float32 band part (rf_f32_band_digits_x10) against the all-f64 sweep: RF gradient and joint gradient of
8192 bench models (random start models and, where gpurun_out/burned_c1.npy exists, burned-in ones), time of pass B alone."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rfsurfhmc_amd._lib import K_NAMES
cfgi = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = bench.CONFIGS[cfgi]
n = cfg["n"]
joint, x_true, bounds = bench.make_joint(cfg, 0)
ctx = joint._ensure(n)
joint.set_warm_start(0)
sets = {"start models": bench.make_models(8192, 991206, n)}
rs = np.random.default_rng(1)
rough = np.clip(x_true[None, :] * (1 + 0.15 * rs.standard_normal((8192, 2 * n))), bounds[:, 0], bounds[:, 1])
sets["unsorted models"] = rough
for tag, xs in sets.items():
    x = torch.from_numpy(xs).cuda()
    out = {}
    for opt in (0, 29, 20):
        ctx.set_option("rf_f32_band_digits_x10", opt)
        m, g, d, f = joint.misfit_and_grad_device(x)
        torch.cuda.synchronize()
        out[opt] = (m.cpu().numpy(), g.cpu().numpy(), d.cpu().numpy(), f.cpu().numpy() != 0)
    ok = out[0][3]
    for opt in (29, 20):
        g0, g1 = out[0][1][ok], out[opt][1][ok]
        rel = np.abs(g1 - g0).max(axis=1) / np.abs(g0).max(axis=1)
        print(f"{tag}: option {opt} vs all-f64: joint gradient max {rel.max():.2e} p99 {np.quantile(rel, 0.99):.2e} median {np.median(rel):.2e}; "
              f"misfit identical {np.array_equal(out[0][0], out[opt][0])}, trace identical {np.array_equal(out[0][2], out[opt][2])}")
# pass B alone
x = torch.from_numpy(sets["start models"]).cuda()
for opt in (0, 29):
    ctx.set_option("rf_f32_band_digits_x10", opt)
    for _ in range(3): joint.misfit_and_grad_device(x)
    ctx.check(ctx.L.rfs_synchronize(ctx.h)); ctx.check(ctx.L.rfs_enable_timing(ctx.h, 1))
    for _ in range(10): joint.misfit_and_grad_device(x)
    ms = np.zeros(len(K_NAMES)); cnt = np.zeros(len(K_NAMES), dtype=np.int32)
    ctx.check(ctx.L.rfs_kernel_ms_sum(ctx.h, ms.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p)))
    ctx.check(ctx.L.rfs_enable_timing(ctx.h, 0))
    print(f"option {opt}: pass B {ms[3] / max(cnt[3], 1):.3f} ms per evaluation (beside the root search), pass A {ms[1] / max(cnt[1], 1):.3f}")
