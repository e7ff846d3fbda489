"""One-off soak: random RF configurations, float32 sweep beyond the band on vs off (not part of the suite)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from rfsurfhmc_amd.model.model_rf import ReceiverFunc
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncfg = int(sys.argv[2]) if len(sys.argv) > 2 else 200
worst = dict(d=0.0, m=0.0, g=0.0); used = resw = refused = 0; bad = []
t0 = time.time()
for it in range(ncfg):
    n = int(rng.integers(2, 61))
    nt = int(rng.choice([128, 200, 256, 400, 512, 700, 1024, 2048]))
    dt = float(rng.choice([0.025, 0.05, 0.1, 0.2]))
    p = float(rng.uniform(0.02, 0.1)); f0 = float(rng.choice([0.5, 1.0, 1.5, 2.5, 4.0]))
    water = float(rng.choice([1e-4, 1e-3, 1e-2, 0.1, 0.5, 0.9])); typ = "S" if rng.random() < 0.3 else "P"
    tshift = float(rng.uniform(1.0, 6.0))
    depth = float(rng.uniform(10, 150))
    nchain = 12
    vs = 1.8 + 3.0 * rng.random((nchain, n))
    if rng.random() < 0.7: vs = np.sort(vs, axis=1)
    thk = depth / n * (0.5 + rng.random((nchain, n))); thk[:, -1] = 0
    xs = np.hstack((vs, thk))
    out = {}
    st = None
    try:
        for opt in (1, 0):
            rf = ReceiverFunc(p, nt, dt, f0, tshift, water, typ, "freq")
            rf.set_obsdata(np.zeros(nt))
            ctx = rf._ensure(n)
            ctx.set_option("rf_f32_beyond_band", opt)
            out[opt] = rf.misfit_and_grad(xs)
            if opt: st = (ctx.stat("rf_f32_chains"), ctx.stat("rf_f32_resweeps"))
    except Exception as e:
        bad.append((it, repr(e)[:100])); continue
    a, b = out[1], out[0]
    ok = np.isfinite(b[2]).all(axis=1) & np.isfinite(b[1]).all(axis=1)
    if not np.array_equal(np.isfinite(a[2]), np.isfinite(b[2])) or not np.array_equal(np.isfinite(a[1]), np.isfinite(b[1])):
        bad.append((it, "finite pattern", n, nt, dt, p, water)); continue
    if not ok.any(): continue
    sc = np.abs(b[2][ok]).max(axis=1, keepdims=True); sg = np.abs(b[1][ok]).max(axis=1, keepdims=True)
    ed = (np.abs(a[2][ok] - b[2][ok]) / np.maximum(sc, 1e-300)).max()
    eg = (np.abs(a[1][ok] - b[1][ok]) / np.maximum(sg, 1e-300)).max()
    em = (np.abs(a[0][ok] - b[0][ok]) / np.maximum(np.abs(b[0][ok]), 1e-300)).max()
    worst["d"] = max(worst["d"], ed); worst["g"] = max(worst["g"], eg); worst["m"] = max(worst["m"], em)
    used += st[0]; resw += st[1]; refused += (nchain - st[0])
    if ed > 1e-11 or eg > 1e-10 or em > 1e-11:
        bad.append((it, "diff", n, nt, dt, round(p, 3), f0, water, typ, ed, eg, em, st))
print("configs", ncfg, "f32 chains", used, "resweeps", resw, "chains kept on f64", refused, "worst", worst, "%.0f s" % (time.time() - t0))
for b in bad[:20]: print("BAD", b)
