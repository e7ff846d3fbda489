"""-m gpu: seeded random configurations of the joint plugin (layer count, data blocks Rc / Rg / Lc / Lg, flat or
spherical earth, RF method freq / time, P or S receiver function, sample counts) against the CPU oracle.  Every case
goes through rfs_joint_setup2 + rfs_joint_misfit_grad with a small batch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


@pytest.mark.parametrize("seed", range(14))
def test_random_joint_configuration(orc, seed):
    from rfsurfhmc_amd.model.model_rf import ReceiverFunc
    from rfsurfhmc_amd.model.model_surf import SurfWD
    from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(3, 26))
    thk = 1.0 + 5.0 * rng.random(n); thk[-1] = 0.0
    vs = np.sort(2.4 + 2.2 * rng.random(n))
    x0 = np.hstack((vs, thk))
    nper = int(rng.integers(3, 14))
    t = np.sort(4.0 + 36.0 * rng.random(nper))
    blocks = dict(tRc=t)
    if rng.random() < 0.5:
        blocks["tRg"] = t
    if rng.random() < 0.4:
        blocks["tLc"] = t
    if rng.random() < 0.3:
        blocks["tLg"] = t
    sphere = bool(rng.random() < 0.4)
    method = "time" if rng.random() < 0.35 else "freq"
    rf_type = "S" if rng.random() < 0.25 else "P"
    nt = int(rng.integers(40, 200)); dt = float(rng.choice([0.1, 0.2, 0.4]))
    rfargs = (0.04 + 0.02 * rng.random(), nt, dt, float(rng.choice([1.0, 1.5, 2.5])), 3.0 + 3.0 * rng.random(), 0.001,
              rf_type, method)
    s1, s2 = 1.0 + rng.random(), 1.0 + rng.random()
    jo = orc.Joint_RF_SWD(s1, s2, orc.ReceiverFunc(*rfargs), orc.SurfWD(sphere=sphere, **blocks))
    jh = Joint_RF_SWD(s1, s2, ReceiverFunc(*rfargs), SurfWD(sphere=sphere, **blocks))
    drf, dswd, flag = jo.forward(x0)
    assert flag
    drf1, dswd1, flag1 = jh.forward(x0)
    assert flag1 and rel(drf1, drf) < 1e-8 and rel(dswd1, dswd) < 2e-6
    jo.set_obsdata(drf, dswd); jh.set_obsdata(drf, dswd)
    nchain = 5
    xs = np.tile(x0, (nchain, 1))
    xs[:, :n] = np.sort(xs[:, :n] * (0.97 + 0.06 * rng.random((nchain, n))), axis=1)
    xs[:, n:2 * n - 1] *= 0.9 + 0.2 * rng.random((nchain, n - 1))
    mh, gh, dh, fh = jh.misfit_and_grad(xs)
    cfg = (n, sorted(blocks), sphere, method, rf_type, nt, dt)
    for i in range(nchain):
        mo, go, do, fo = jo.misfit_and_grad(xs[i])
        assert fo == bool(fh[i]), cfg
        if not fo:
            continue
        assert rel(dh[i], do) < 2e-6, cfg
        assert abs(mh[i] - mo) <= 1e-5 * max(mo, 1e-12), (cfg, mh[i], mo)
        # group-velocity kernels difference two phase kernels 10 % apart in period: 2e-5 (see test_gpu_parity.py)
        assert rel(gh[i], go) < 2e-5, (cfg, rel(gh[i], go))
