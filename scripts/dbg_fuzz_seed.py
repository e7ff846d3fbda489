"""Debug helper: one seed of tests/test_gpu_fuzz.py with per-block differences printed."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from oracle import oracle as orc
orc.build(ref=False)
from rfsurfhmc_amd.model.model_rf import ReceiverFunc
from rfsurfhmc_amd.model.model_surf import SurfWD
from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
np.set_printoptions(precision=12, linewidth=200)
def rel(a, b): return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))
for seed in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(3, 26))
    thk = 1.0 + 5.0 * rng.random(n); thk[-1] = 0.0
    vs = np.sort(2.4 + 2.2 * rng.random(n))
    x0 = np.hstack((vs, thk))
    nper = int(rng.integers(3, 14))
    t = np.sort(4.0 + 36.0 * rng.random(nper))
    blocks = dict(tRc=t)
    if rng.random() < 0.5: blocks["tRg"] = t
    if rng.random() < 0.4: blocks["tLc"] = t
    if rng.random() < 0.3: blocks["tLg"] = t
    sphere = bool(rng.random() < 0.4)
    method = "time" if rng.random() < 0.35 else "freq"
    rf_type = "S" if rng.random() < 0.25 else "P"
    nt = int(rng.integers(40, 200)); dt = float(rng.choice([0.1, 0.2, 0.4]))
    rfargs = (0.04 + 0.02 * rng.random(), nt, dt, float(rng.choice([1.0, 1.5, 2.5])), 3.0 + 3.0 * rng.random(), 0.001, rf_type, method)
    s1, s2 = 1.0 + rng.random(), 1.0 + rng.random()
    jo = orc.Joint_RF_SWD(s1, s2, orc.ReceiverFunc(*rfargs), orc.SurfWD(sphere=sphere, **blocks))
    jh = Joint_RF_SWD(s1, s2, ReceiverFunc(*rfargs), SurfWD(sphere=sphere, **blocks))
    drf, dswd, flag = jo.forward(x0)
    drf1, dswd1, flag1 = jh.forward(x0)
    print("seed", seed, "n", n, sorted(blocks), "sphere", sphere, rfargs)
    print(" forward flags oracle/hip", flag, flag1)
    if not flag:
        print("  oracle dswd", dswd); print("  hip    dswd", dswd1); continue
    print(" forward rel rf", rel(drf1, drf), "swd", rel(dswd1, dswd))
    jo.set_obsdata(drf, dswd); jh.set_obsdata(drf, dswd)
    nchain = 5
    xs = np.tile(x0, (nchain, 1))
    xs[:, :n] = np.sort(xs[:, :n] * (0.97 + 0.06 * rng.random((nchain, n))), axis=1)
    xs[:, n:2 * n - 1] *= 0.9 + 0.2 * rng.random((nchain, n - 1))
    mh, gh, dh, fh = jh.misfit_and_grad(xs)
    for i in range(nchain):
        mo, go, do, fo = jo.misfit_and_grad(xs[i])
        print("  chain", i, fo, bool(fh[i]), "misfit", mo, mh[i], abs(mh[i] - mo) / mo, "rf", rel(dh[i][:nt], do[:nt]), "swd", rel(dh[i][nt:], do[nt:]), "grad", rel(gh[i], go))
        if abs(mh[i] - mo) > 5e-6 * mo:
            print("   swd oracle", do[nt:]); print("   swd hip   ", dh[i][nt:]); print("   dobs      ", dswd)
