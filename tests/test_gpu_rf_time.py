"""-m gpu: time-domain receiver functions (method="time": iterative deconvolution, the reference's default)
through the C ABI against the CPU oracle.

Pin status (see oracle/oracle.py): the oracle's per-frequency R21 / R22 / partials are pinned to the compiled
reference core; its deconit is a numpy restatement of src/RF/deconit.f90 that cannot be pinned here (FFTW3 is
absent), so these tests establish HIP == restatement, not HIP == reference, for the deconvolution itself.
The device algorithm is NOT the restatement's (no FFT inside the loop, see rf_time_kernels.hpp), which makes the
agreement below a meaningful cross-check of both."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


YAML7 = (np.array([6., 6, 13., 5, 10, 30, 0]), np.array([3.2, 2.8, 3.46, 3.3, 3.9, 4.5, 4.7]))


@pytest.fixture(scope="module")
def hip():
    from rfsurfhmc_amd.model.lib import librf
    from rfsurfhmc_amd.model.model_rf import ReceiverFunc
    from rfsurfhmc_amd.model.model_surf import SurfWD
    from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD

    class H:
        pass
    h = H()
    h.librf, h.ReceiverFunc, h.SurfWD, h.Joint = librf, ReceiverFunc, SurfWD, Joint_RF_SWD
    return h


@pytest.mark.parametrize("nt,dt,rf_type", [(125, 0.4, "P"), (256, 0.2, "P"), (100, 0.4, "S"), (60, 0.5, "P")])
def test_librf_forward_time(hip, orc, nt, dt, rf_type):
    thk, vs = YAML7
    vp, rho, _, _ = orc.empirical_relation(vs)
    q = np.full(len(vs), 9999.)
    ref = orc.librf.forward(thk, rho, vp, vs, q, q, 0.045, nt, dt, 1.5, 5.0, "time", 0.001, rf_type)
    got = hip.librf.forward(thk, rho, vp, vs, q, q, 0.045, nt, dt, 1.5, 5.0, "time", 0.001, rf_type)
    assert got.shape == (nt,)
    assert rel(got, ref) < 1e-9, rel(got, ref)          # observed ~1e-13: same spikes, same amplitudes
    # default method of the binding is "time" (src/RF/main.cpp:21)
    assert np.array_equal(got, hip.librf.forward(thk, rho, vp, vs, q, q, 0.045, nt, dt, 1.5, 5.0, rf_type=rf_type))


def test_librf_kernel_all_time(hip, orc):
    thk, vs = YAML7
    vp, rho, _, _ = orc.empirical_relation(vs)
    q = np.full(len(vs), 9999.)
    rf0, kl0 = orc.librf.kernel_all(thk, rho, vp, vs, q, q, 0.045, 125, 0.4, 1.5, 5.0, "time", 0.001, "P")
    rf1, kl1 = hip.librf.kernel_all(thk, rho, vp, vs, q, q, 0.045, 125, 0.4, 1.5, 5.0, "time", 0.001, "P")
    assert rel(rf1, rf0) < 1e-9
    # up to 200 greedy spikes per trace, every one the same on both sides: observed <= 1e-14 per trace
    for ip in range(4):
        for j in range(len(vs)):
            assert rel(kl1[ip, j], kl0[ip, j]) < 1e-8 or not np.any(kl0[ip, j]), (ip, j)
    assert not np.any(kl1[3, -1])                        # half-space thickness: zero trace
    # single-parameter entry: cal_rf_par_time's frequency axis (float32 pi)
    rf2, k2 = hip.librf.kernel(thk, rho, vp, vs, q, q, 0.045, 125, 0.4, 1.5, 5.0, "time", 0.001, "P", "vs")
    rf3, k3 = orc.librf.kernel(thk, rho, vp, vs, q, q, 0.045, 125, 0.4, 1.5, 5.0, "time", 0.001, "P", "vs")
    assert rel(rf2, rf3) < 1e-9 and rel(k2, k3) < 1e-8


def test_time_batched_equals_single_and_30_layers(hip, orc):
    rng = np.random.default_rng(8)
    n, nchain = 30, 9
    thk = np.full(n, 2.0); thk[-1] = 0
    vs = np.linspace(2.8, 4.6, n) * (0.97 + 0.06 * rng.random((nchain, n)))
    thk = np.tile(thk, (nchain, 1))
    vp, rho, _, _ = orc.empirical_relation(vs)
    q = np.full((nchain, n), 9999.)
    rfb = hip.librf.forward(thk, rho, vp, vs, q, q, 0.045, 512, 0.1, 1.5, 5.0, "time")
    for i in (0, 4, 8):
        one = hip.librf.forward(thk[i], rho[i], vp[i], vs[i], q[i], q[i], 0.045, 512, 0.1, 1.5, 5.0, "time")
        assert np.array_equal(one, rfb[i])
        ref = orc.librf.forward(thk[i], rho[i], vp[i], vs[i], q[i], q[i], 0.045, 512, 0.1, 1.5, 5.0, "time")
        assert rel(one, ref) < 1e-9


def test_plugin_time_misfit_and_grad(hip, orc):
    """ReceiverFunc(method="time").misfit_and_grad: the spike-wise gradient (no kernel traces on the device)
    against kernel_all @ residual of the oracle; joint with Rc data as well."""
    thk, vs = YAML7
    n = len(vs)
    x0 = np.hstack((vs, thk))
    rng = np.random.default_rng(21)
    xs = np.tile(x0, (6, 1)); xs[:, :n] *= 0.98 + 0.04 * rng.random((6, n))
    args = (0.045, 125, 0.4, 1.5, 5.0, 0.001, "P", "time")
    mo = orc.ReceiverFunc(*args); mh = hip.ReceiverFunc(*args)
    d0 = mo.forward(x0)
    assert rel(mh.forward(x0), d0) < 1e-9
    mo.set_obsdata(d0); mh.set_obsdata(d0)
    mfh, gh, dh = mh.misfit_and_grad(xs)
    for i in range(6):
        mf, g, d = mo.misfit_and_grad(xs[i])
        assert rel(dh[i], d) < 1e-9
        assert abs(mfh[i] - mf) <= 1e-8 * mf
        assert rel(gh[i], g) < 1e-8, (i, rel(gh[i], g))
    t = np.linspace(5, 40, 12)
    jo = orc.Joint_RF_SWD(1.0, 1.0, orc.ReceiverFunc(*args), orc.SurfWD(tRc=t))
    jh = hip.Joint(1.0, 1.0, hip.ReceiverFunc(*args), hip.SurfWD(tRc=t))
    drf, dswd, flag = jo.forward(x0)
    drf1, dswd1, flag1 = jh.forward(x0)
    assert flag and flag1 and rel(drf1, drf) < 1e-9 and rel(dswd1, dswd) < 2e-6
    jo.set_obsdata(drf, dswd); jh.set_obsdata(drf, dswd)
    m1, g1, d1, f1 = jh.misfit_and_grad(xs)
    for i in range(6):
        m0, g0, d0_, f0 = jo.misfit_and_grad(xs[i])
        assert f0 == bool(f1[i]) and rel(d1[i], d0_) < 2e-6
        assert abs(m1[i] - m0) <= 1e-5 * m0 and rel(g1[i], g0) < 1e-5


@pytest.mark.parametrize("nt,dt", [(10, 1.0), (33, 0.5), (100, 0.4), (1000, 0.05), (2048, 0.025), (4096, 0.0125),
                                   (5000, 0.01), (9000, 0.006)])
def test_time_domain_length_extremes(hip, orc, nt, dt):
    """FFT lengths 16 (fewer lags than a wavefront), 64, 128 (exactly one lag per lane), 1024 (8 lags per lane),
    2048 and 4096 (the longest trace one wavefront holds: 32 lags per lane, > 64 KB of LDS per block), and 8192 /
    16384 (one block per trace, lags in place: k_rft_deconv_big -- the reference has no length limit, deconit.f90)."""
    thk, vs = YAML7
    vp, rho, _, _ = orc.empirical_relation(vs)
    q = np.full(len(vs), 9999.)
    args = (thk, rho, vp, vs, q, q, 0.045, nt, dt, 1.5, 3.0, "time", 0.001, "P")
    rf0, kl0 = orc.librf.kernel_all(*args)
    rf1, kl1 = hip.librf.kernel_all(*args)
    assert rel(rf1, rf0) < 1e-8
    for ip in range(4):
        for j in range(len(vs)):
            assert rel(kl1[ip, j], kl0[ip, j]) < 1e-7 or not np.any(kl0[ip, j]), (ip, j, rel(kl1[ip, j], kl0[ip, j]))


def test_plugin_time_long_trace(hip, orc):
    """B2 with method "time" and nt = 5000 (8192-point transforms): the block-per-trace deconvolution, the L2-read
    pulse and the tiled residual correlation behind misfit_and_grad."""
    thk, vs = YAML7
    n = len(vs)
    x0 = np.hstack((vs, thk))
    xs = np.tile(x0, (2, 1)); xs[1, :n] *= 1.015
    args = (0.045, 5000, 0.01, 1.5, 5.0, 0.001, "P", "time")
    mo = orc.ReceiverFunc(*args); mh = hip.ReceiverFunc(*args)
    d0 = mo.forward(x0)
    assert rel(mh.forward(x0), d0) < 1e-8
    mo.set_obsdata(d0); mh.set_obsdata(d0)
    mfh, gh, dh = mh.misfit_and_grad(xs)
    mf, g, d = mo.misfit_and_grad(xs[1])
    assert rel(dh[1], d) < 1e-8 and abs(mfh[1] - mf) <= 1e-8 * mf and rel(gh[1], g) < 1e-7, (rel(dh[1], d), rel(gh[1], g))


def test_time_domain_two_layers_and_many_layers(hip, orc):
    q2 = np.full(2, 9999.)
    thk, vs = np.array([20.0, 0.0]), np.array([3.2, 4.4])
    vp, rho, _, _ = orc.empirical_relation(vs)
    a = (thk, rho, vp, vs, q2, q2, 0.05, 64, 0.4, 1.5, 3.0, "time", 0.001, "P")
    assert rel(hip.librf.kernel_all(*a)[1], orc.librf.kernel_all(*a)[1]) < 1e-7
    n = 40
    thk = np.full(n, 1.5); thk[-1] = 0; vs = np.linspace(2.6, 4.6, n)
    vp, rho, _, _ = orc.empirical_relation(vs); q = np.full(n, 9999.)
    a = (thk, rho, vp, vs, q, q, 0.045, 128, 0.2, 1.5, 5.0, "time", 0.001, "P")
    assert rel(hip.librf.forward(*a), orc.librf.forward(*a)) < 1e-8


def test_reference_smoke_script_configuration(hip, orc):
    """The reference's only runnable check, test_forward.py:13-45: Joint_RF_SWD.forward on its 7-layer model with a
    time-domain P receiver function (nt = 500 -> 512-point transforms, gauss 1.0) and 36 Rc + 36 Rg periods
    (the script only plots; here the same call is compared with the oracle, whose SWD part is the compiled reference)."""
    tRc = np.linspace(5, 40, 36)
    args = (0.045, 500, 0.1, 1.0, 5.0, 0.001, "P", "time")
    thk = np.array([6., 6, 13, 5, 10, 30, 0])
    vs = np.array([3.2, 3.4, 3.46, 3.7, 3.9, 4.5, 4.7])
    x = np.hstack((vs, thk))
    mo_rf, mo_swd = orc.ReceiverFunc(*args), orc.SurfWD(tRc=tRc, tRg=tRc.copy())
    mh_rf, mh_swd = hip.ReceiverFunc(*args), hip.SurfWD(tRc=tRc, tRg=tRc.copy())
    for m in (mh_rf, mh_swd):
        m.set_thk(thk)                                                    # test_forward.py:33-34
    jo = orc.Joint_RF_SWD(1.0, 1.0, mo_rf, mo_swd)
    jh = hip.Joint(1.0, 1.0, mh_rf, mh_swd)
    dr0, ds0, f0 = jo.forward(x)
    dr1, ds1, f1 = jh.forward(x)
    assert f0 and f1 and dr1.shape == (500,) and ds1.shape == (72,)
    assert rel(dr1, dr0) < 1e-9
    assert np.array_equal(ds1[:36], ds0[:36])                             # float32-rounded phase velocities
    assert rel(ds1[36:], ds0[36:]) < 1e-6
    assert abs(np.argmax(dr1) * 0.1 - 5.0) < 0.15                         # direct P at the time shift
