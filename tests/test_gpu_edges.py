"""-m gpu: shapes at the edges of the C ABI (smallest / largest models, tiny and odd FFT lengths, single
periods, every root-search kernel variant, argument errors) against the CPU oracle."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


def _model(n, rng, sort=True):
    vs = 2.2 + 2.4 * rng.random(n)
    if sort:
        vs = np.sort(vs)
    thk = 0.8 + 3.0 * rng.random(n); thk[-1] = 0.0
    return vs, thk


@pytest.mark.parametrize("n", [2, 3, 33, 64, 65, 128])
def test_layer_count_extremes_b2(orc, n):
    from rfsurfhmc_amd.model.model_rf import ReceiverFunc
    from rfsurfhmc_amd.model.model_surf import SurfWD
    from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
    rng = np.random.default_rng(n)
    vs, thk = _model(n, rng)
    if n > 40:
        thk[:-1] *= 40.0 / n            # keep the stack ~60 km thick
    t = np.array([6.0, 11.0, 23.0])
    nt = 100
    j = Joint_RF_SWD(1.0, 2.0, ReceiverFunc(0.05, nt, 0.25, 2.0, 3.0, 0.01, "P", "freq"), SurfWD(tRc=t, tRg=t))
    o = orc.Joint_RF_SWD(1.0, 2.0, orc.ReceiverFunc(0.05, nt, 0.25, 2.0, 3.0, 0.01, "P", "freq"), orc.SurfWD(tRc=t, tRg=t))
    x0 = np.hstack((vs, thk))
    drf, dswd, fl = o.forward(x0)
    assert fl
    j.set_obsdata(drf, dswd); o.set_obsdata(drf, dswd)
    x = x0 * (1 + 0.02 * (rng.random(2 * n) - 0.5))
    m, g, d, f = j.misfit_and_grad(x)
    mo, go, do, fo = o.misfit_and_grad(x)
    assert f == fo
    assert rel(d, do) < 1e-6 and abs(m - mo) <= 1e-5 * abs(mo) and rel(g, go) < 1e-5, (n, rel(g, go))


@pytest.mark.parametrize("nt,dt", [(3, 1.0), (16, 0.5), (33, 0.5), (64, 0.4), (127, 0.3), (129, 0.3), (1000, 0.05)])
def test_fft_length_extremes_b1(orc, nt, dt):
    """nft = 4 ... 1024: fewer frequencies than a wavefront, one partial block, several blocks."""
    from rfsurfhmc_amd.model.lib import librf
    rng = np.random.default_rng(nt)
    vs, thk = _model(6, rng)
    vp, rho, _, _ = orc.empirical_relation(vs)
    q = np.full(6, 9999.)
    args = (thk, rho, vp, vs, q, q, 0.06, nt, dt, 1.2, 2.0, "freq", 0.005, "P")
    rf, kl = librf.kernel_all(*args)
    rfo, klo = orc.librf.kernel_all(*args)
    assert rf.shape == (nt,) and kl.shape == (4, 6, nt)
    assert rel(rf, rfo) < 1e-9 and rel(kl, klo) < 1e-8
    assert rel(librf.forward(*args), orc.librf.forward(*args)) < 1e-9


def _root_search_problem(orc, nchain, n=12, seed=77):
    rng = np.random.default_rng(seed)
    vs = np.sort(2.0 + 2.6 * rng.random((nchain, n)), axis=1)
    vs[5] = vs[5, ::-1]                       # a velocity-inversion model
    thk = 1.0 + 3 * rng.random((nchain, n)); thk[:, -1] = 0
    vp, rho, _, _ = orc.empirical_relation(vs)
    return thk, vp, vs, rho, np.linspace(3, 35, 7)


def _roots(nchain, n, model, t, **options):
    from rfsurfhmc_amd._lib import Context, hptr
    ctx = Context(0, max_chains=4096)
    for k, v in options.items():
        ctx.check(ctx.L.rfs_set_option(ctx.h, k.encode(), v))
    c = np.zeros((nchain, len(t))); flag = np.zeros(nchain, dtype=np.int32)
    a = [np.ascontiguousarray(v) for v in model]
    ctx.check(ctx.L.rfs_swd_forward(ctx.h, nchain, n, hptr(a[0]), hptr(a[1]), hptr(a[2]), hptr(a[3]), len(t), hptr(t),
                                    0, 0, 0, hptr(c), hptr(flag)))
    ctx.close()
    return c, flag


# lanes = 0 is the automatic choice: <= 3072 items the latency form of the lanes-per-item kernel (one / two / four items
# per wavefront, recurrence in four segments, scan look-ahead), above it the cooperative producer / consumer blocks
@pytest.mark.parametrize("lanes,nchain", [(1, 1100), (2, 1100), (4, 1100), (8, 1100), (16, 1100), (32, 1100), (64, 1100),
                                          (0, 3500), (0, 1500), (0, 1100), (0, 200)])
def test_every_root_search_kernel_agrees(orc, lanes, nchain):
    """lane-per-chain, G-lanes-per-chain (LDS; plain and latency form) and cooperative kernels return the reference's
    roots and flags."""
    n = 12
    thk, vp, vs, rho, t = _root_search_problem(orc, nchain, n)
    c, flag = _roots(nchain, n, (thk, vp, vs, rho), t, swd_lanes_per_chain=lanes)
    for i in list(range(0, nchain, 97)) + [5]:
        co, fo = orc.libsurf.forward(thk[i], vp[i], vs[i], rho[i], t, "Rc")
        assert bool(flag[i]) == fo
        assert np.all(np.abs(c[i] - co) <= 1.2e-6 * np.abs(co) + 1e-300), (lanes, i)


def test_scan_look_ahead_changes_nothing_and_segments_only_rounding(orc):
    """Speculation feeds the state machine exactly the (request, Delta) pairs of the one-at-a-time search: bit-identical
    roots for every chain.  The segmented recurrence is the same product associated differently: roots equal to rounding
    (a float32-rounded root may land on the neighbouring value once in thousands)."""
    n, nchain = 12, 900
    thk, vp, vs, rho, t = _root_search_problem(orc, nchain, n, seed=78)
    mdl = (thk, vp, vs, rho)
    base, fb = _roots(nchain, n, mdl, t, swd_lanes_per_chain=32, swd_segments=1, swd_speculate=1)
    seq, fs = _roots(nchain, n, mdl, t, swd_lanes_per_chain=1)
    assert np.array_equal(base, seq) and np.array_equal(fb, fs)
    for spec in (2, 4):
        for lanes in (16, 64):
            c, f = _roots(nchain, n, mdl, t, swd_lanes_per_chain=lanes, swd_segments=1, swd_speculate=spec)
            assert np.array_equal(c, base) and np.array_equal(f, fb), (spec, lanes)
    for seg, lanes in ((2, 8), (4, 16), (4, 64)):
        c, f = _roots(nchain, n, mdl, t, swd_lanes_per_chain=lanes, swd_segments=seg, swd_speculate=4)
        assert np.array_equal(f, fb)
        ndiff = int((c != base).sum())
        assert ndiff <= 3 and np.all(np.abs(c - base) <= 1.2e-6 * np.abs(base)), (seg, lanes, ndiff)
    auto, fa = _roots(nchain, n, mdl, t)
    assert np.array_equal(fa, fb) and int((auto != base).sum()) <= 3
    # nothing fancy is accepted silently
    from rfsurfhmc_amd._lib import Context
    ctx = Context(0, max_chains=8)
    assert ctx.L.rfs_set_option(ctx.h, b"swd_speculate", 3) == -1 and ctx.L.rfs_set_option(ctx.h, b"swd_segments", 8) == -1
    ctx.close()


def test_single_period_and_group_only(orc):
    from rfsurfhmc_amd.model.model_surf import SurfWD
    rng = np.random.default_rng(5)
    vs, thk = _model(9, rng)
    x = np.hstack((vs, thk))
    for kw in (dict(tRc=[17.0]), dict(tRg=[8.0, 21.0]), dict(tRc=[9.0], tRg=[9.0])):
        s = SurfWD(**kw); o = orc.SurfWD(**kw)
        d0, fl = o.forward(x)
        dobs = d0 * 1.01
        s.set_obsdata(dobs); o.set_obsdata(dobs)
        m, g, d, f = s.misfit_and_grad(x)
        mo, go, do, fo = o.misfit_and_grad(x)
        assert f and fo and rel(d, do) < 1e-6 and abs(m - mo) < 1e-5 * mo and rel(g, go) < 1e-5, kw


def test_argument_errors_are_reported_not_fatal():
    from rfsurfhmc_amd._lib import Context, RfParams, RfsError, hptr
    ctx = Context(0, max_chains=8, max_layers=16)
    L = ctx.L
    x = np.zeros((1, 8)); out = np.zeros(64); fl = np.zeros(1, dtype=np.int32)
    assert L.rfs_joint_misfit_grad(ctx.h, 1, hptr(x), hptr(out), hptr(out), hptr(out), hptr(fl)) == -3     # no setup yet
    t = np.array([5.0, 10.0])
    assert L.rfs_joint_setup(ctx.h, 40, None, 2, hptr(t), 0, None, 1.0, 1.0, None) == -1                    # > max_layers
    assert L.rfs_joint_setup(ctx.h, 4, None, 0, None, 0, None, 1.0, 1.0, None) == -1                        # no data at all
    par = RfParams(0.05, 64, 0.2, 1.5, 2.0, 0.001, 1, 5)                                                    # unknown method
    assert L.rfs_joint_setup(ctx.h, 4, ctypes.byref(par), 0, None, 0, None, 1.0, 1.0, None) == -1
    par = RfParams(0.05, 64, 0.2, 1.5, 2.0, 0.001, 1, 0)
    par.method, par.rf_type = 1, 7
    assert L.rfs_joint_setup(ctx.h, 4, ctypes.byref(par), 0, None, 0, None, 1.0, 1.0, None) == -1
    a = np.ones((1, 4))
    assert L.rfs_swd_forward(ctx.h, 1, 4, hptr(a), hptr(a), hptr(a), hptr(a), 2, hptr(t), 2, -1, 0, hptr(out), hptr(fl)) == -1   # bad mode
    assert L.rfs_swd_forward(ctx.h, 1, 4, hptr(a), hptr(a), hptr(a), hptr(a), 2, hptr(t), 7, 0, 0, hptr(out), hptr(fl)) == -1   # bad wavetype
    assert L.rfs_swd_forward(ctx.h, 9, 4, hptr(a), hptr(a), hptr(a), hptr(a), 2, hptr(t), 0, 0, 0, hptr(out), hptr(fl)) == -1   # > max_chains
    assert b"max_chains" in L.rfs_last_error(ctx.h)
    # newer entries: state and argument errors
    one = np.ones(8)
    assert L.rfs_set_inverse_mass(ctx.h, hptr(one)) == -3                                                   # before any setup
    dummy = ctypes.c_void_p(8)          # never dereferenced: the calls below fail before touching device memory
    assert L.rfs_flow_step(ctx.h, 1, *([dummy] * 14)) == -3
    assert L.rfs_joint_setup(ctx.h, 4, None, 2, hptr(t), 0, None, 1.0, 1.0, None) == 0
    assert L.rfs_set_inverse_mass(ctx.h, hptr(np.r_[np.ones(7), -1.0])) == -1                               # non-positive mass
    assert L.rfs_set_inverse_mass(ctx.h, hptr(one)) == 0 and L.rfs_set_inverse_mass(ctx.h, None) == 0
    bad = np.array([1, 2], dtype=np.int32)                                                                  # nactive must not increase
    assert L.rfs_leapfrog_dev2(ctx.h, 2, dummy, dummy, dummy, dummy, 2, hptr(bad), *([dummy] * 9)) == -1
    assert L.rfs_flow_step(ctx.h, 1, dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy,
                           dummy, dummy, None) == -1                                                        # null output
    from rfsurfhmc_amd._lib import FlowNext
    nxt = FlowNext()                                                                                        # all null
    assert L.rfs_flow_step2(ctx.h, 1, *([dummy] * 14), ctypes.byref(nxt)) == -1 and b"rfs_flow_next" in L.rfs_last_error(ctx.h)
    assert L.rfs_set_option(ctx.h, b"early_eigen_periods", -2) == -1 and L.rfs_set_option(ctx.h, b"no_such_option", 1) == -1
    assert L.rfs_set_option(ctx.h, b"early_eigen_periods", 0) == 0 and L.rfs_set_option(ctx.h, b"early_eigen_periods", -1) == 0
    with pytest.raises(RfsError):
        ctx.check(-1)
    ctx.close()


def test_schedules_give_identical_results(orc):
    """The joint evaluation on a caller-provided stream in its two schedules -- root search and RF kernels sharing the
    CUs, or on disjoint halves of the CU mask -- returns the same misfit, gradient, synthetics and flags bit for bit
    (Rc + Rg data, one velocity-inversion chain)."""
    import torch
    from rfsurfhmc_amd.model.model_rf import ReceiverFunc
    from rfsurfhmc_amd.model.model_surf import SurfWD
    from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
    rng = np.random.default_rng(12)
    nchain, n = 1536, 14
    vs = np.sort(2.2 + 2.4 * rng.random((nchain, n)), axis=1)
    vs[7] = vs[7, ::-1]
    thk = 1.0 + 3 * rng.random((nchain, n)); thk[:, -1] = 0
    x = np.hstack((vs, thk))
    t = np.linspace(4, 36, 17)
    joint = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(0.045, 128, 0.2, 1.5, 5.0, 0.001, "P", "freq"), SurfWD(tRc=t, tRg=t))
    drf, dswd, flag = joint.forward(np.hstack((np.linspace(2.5, 4.5, n), np.r_[np.full(n - 1, 2.5), 0])))
    assert flag
    joint.set_obsdata(drf, dswd)
    xd = torch.from_numpy(x).cuda()
    ctx = joint._ensure(n)
    res = {}
    for name, cu in (("shared", 0), ("partitioned", 1), ("partitioned2", 2)):
        ctx.check(ctx.L.rfs_set_option(ctx.h, b"cu_split", cu))
        with torch.cuda.stream(torch.cuda.Stream()):
            out = joint.misfit_and_grad_device(xd)
            torch.cuda.current_stream().synchronize()
        ctx.check(ctx.L.rfs_synchronize(ctx.h))
        res[name] = [o.cpu().numpy().copy() for o in out]
    assert res["shared"][3].sum() >= nchain - 8 and np.all(np.isfinite(res["shared"][1]))
    for name in ("partitioned", "partitioned2"):
        for a, b in zip(res["shared"], res[name]):
            assert np.array_equal(a, b), name


def test_plain_c_client_of_the_abi(tmp_path):
    """examples/c_client.c, compiled with gcc against librfsurf_hip.so, reproduces the values observed from the
    reference on its param.yaml model."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "rfsurfhmc_amd")
    exe = str(tmp_path / "c_client")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", f"-I{root}", os.path.join(root, "examples", "c_client.c"),
                    "-o", exe, f"-L{libdir}", "-l:librfsurf_hip.so", f"-Wl,-rpath,{libdir}", "-lm"], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.splitlines()
    assert out[0] == "flag 1"
    c = np.array(out[1].split()[1:], dtype=float)
    np.testing.assert_allclose(c, [2.811252593994, 2.802712202072, 2.807658672333, 2.823679924011, 2.847761392593,
                                   2.876952648163], rtol=0, atol=5e-13)
    kb = np.array(out[2].split()[1:], dtype=float)
    np.testing.assert_allclose(kb, [4.180354642102e-01, 3.241766678501e-01, 3.022816772550e-02, 7.717666538807e-05,
                                    3.931093706971e-06, 7.434082888798e-09, 1.085835992717e-17], rtol=2e-6)
    rf = np.array(out[3].split()[1:], dtype=float)
    np.testing.assert_allclose(rf, [1.750384485169e-04, 2.704993610476e-03, 2.454749574455e-02, 1.064170021474e-01,
                                    2.228427016323e-01, 2.188586786966e-01, 9.057466370315e-02, 1.807677009266e-02], rtol=1e-9)


def test_rg_block_shares_the_rc_blocks_pass_when_periods_coincide(orc):
    """tRg == tRc (param.yaml's own set-up): sregnpu's pass at T is the Rc block's search and eigenfunction pass, so the Rg
    block reads those items (option share_rc_rg, default on).  Same numbers bit for bit as computing it again, and as
    close to the oracle as before; different period lists are not shared."""
    from rfsurfhmc_amd.model.model_surf import SurfWD
    rng = np.random.default_rng(12)
    vs, thk = _model(9, rng)
    x0 = np.hstack((vs, thk))
    xs = np.tile(x0, (6, 1)); xs[:, :9] = np.sort(xs[:, :9] * (0.98 + 0.04 * rng.random((6, 9))), axis=1)
    t = np.linspace(6.0, 30.0, 7)

    def run(share, tRg, love=False):
        m = SurfWD(tRc=t, tRg=tRg, tLc=t, tLg=t) if love else SurfWD(tRc=t, tRg=tRg)
        d0, fl = m.forward(x0)
        m.set_obsdata(d0 * 1.01)
        ctx = m._ensure(9)
        ctx.check(ctx.L.rfs_set_option(ctx.h, b"share_rc_rg", share))
        m.set_obsdata(d0 * 1.01)                       # reconfigures the context with the option in force
        out = m.misfit_and_grad(xs)
        fwd = m.forward(xs[0])
        return d0, out, fwd
    d_on, on, f_on = run(1, t)
    d_off, off, f_off = run(0, t)
    assert np.array_equal(d_on, d_off) and np.array_equal(f_on[0], f_off[0])
    for a, b in zip(on, off):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    o = orc.SurfWD(tRc=t, tRg=t); o.set_obsdata(d_on * 1.01)
    for i in range(6):
        mo, go, do, fo = o.misfit_and_grad(xs[i])
        assert fo and rel(on[2][i], do) < 2e-6 and abs(on[0][i] - mo) <= 1e-5 * mo and rel(on[1][i], go) < 1e-4
    # all four blocks on the same periods: both group blocks share
    d4, o4, f4 = run(1, t, love=True)
    d4n, o4n, f4n = run(0, t, love=True)
    assert np.array_equal(d4, d4n) and np.array_equal(f4[0], f4n[0])
    for a, b in zip(o4, o4n):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    # periods that differ in one entry: nothing to share, results still agree with the oracle
    t2 = t.copy(); t2[3] += 0.25
    d2, o2, _ = run(1, t2)
    oo = orc.SurfWD(tRc=t, tRg=t2); oo.set_obsdata(d2 * 1.01)
    mo, go, do, fo = oo.misfit_and_grad(xs[0])
    assert fo and rel(o2[2][0], do) < 2e-6 and rel(o2[1][0], go) < 1e-4


@pytest.mark.parametrize("n", [2, 3, 5, 9, 17, 33, 65, 128])
def test_layer_count_extremes_love_search(orc, n):
    """The Love root search through the lanes-per-item kernel at layer counts around its lane / segment boundaries
    (1 layer over a half-space, fewer layers than segments, exactly one more than a power of two, the maximum):
    SurfWD with all four blocks, flat and spherical, against the oracle."""
    from rfsurfhmc_amd.model.model_surf import SurfWD
    rng = np.random.default_rng(100 + n)
    vs, thk = _model(n, rng)
    if n > 40:
        thk[:-1] *= 40.0 / n
    t = np.array([6.0, 11.0, 23.0])
    x0 = np.hstack((vs, thk))
    for sph in (False, True):
        kw = dict(tRc=t, tRg=t, tLc=t, tLg=t, sphere=sph)
        s, o = SurfWD(**kw), orc.SurfWD(**kw)
        d0, fl = o.forward(x0)
        assert fl
        s.set_obsdata(d0 * 1.01); o.set_obsdata(d0 * 1.01)
        x = x0 * (1 + 0.02 * (rng.random(2 * n) - 0.5))
        m, g, d, f = s.misfit_and_grad(x)
        mo, go, do, fo = o.misfit_and_grad(x)
        assert f == fo and f
        assert rel(d, do) < 2e-6 and abs(m - mo) <= 1e-5 * abs(mo) and rel(g, go) < 1e-4, (n, sph, rel(d, do), rel(g, go))


def test_flow_step2_deferred_form_equals_the_plain_step(orc):
    """rfs_flow_step2 with gsave / kick (the dual-averaging form): every chain's first half kick is applied one call
    later, from the saved gradient -- positions, momenta, energies and synthetics after a few calls equal those of
    rfs_flow_step bit for bit, also across a restart made by the device with the step size written one call late."""
    import torch
    from rfsurfhmc_amd.model.model_surf import SurfWD
    rng = np.random.default_rng(8)
    vs, thk = _model(8, rng)
    x0 = np.hstack((vs, thk))
    t = np.linspace(6.0, 30.0, 6)
    m = SurfWD(tRc=t)
    m.set_warm_start(0)          # two flow states share this context call by call: each call must stand on its own
    d0, fl = m.forward(x0); m.set_obsdata(d0 * 1.01)
    nc, nx = 48, 16
    xs = np.tile(x0, (nc, 1)) * (1 + 0.01 * rng.standard_normal((nc, nx)))
    xs[:, :8] = np.sort(xs[:, :8], axis=1)
    lo, hi = xs.min(0) * 0.8, xs.max(0) * 1.2
    bounds = np.stack([lo, hi], axis=1)
    dev = torch.device("cuda")
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    p0 = 0.5 * rng.standard_normal((nc, nx)); p1 = 0.5 * rng.standard_normal((nc, nx))
    dt0 = np.full(nc, 0.01); dt1 = 0.01 * (1 + 0.3 * rng.random(nc)); L0, L1 = 2, 3

    def start(deferred):
        st = m.flow_state(tt(xs), tt(dt0), tt(bounds))
        st["p"].copy_(tt(p0)); st["rem"].fill_(L0); st["fresh"].fill_(1)
        return m.flow_restart_state(st, deferred=True) if deferred else st
    a, b = start(False), start(True)
    for s in range(L0 + 1):                     # first trajectory: start evaluation + L0 steps
        if s == L0:
            b["nxt_u"].fill_(0.0); b["nxt_p"].copy_(tt(p1)); b["nxt_have"].fill_(1)      # u = 0: certain accept
        m.flow_step(a); m.flow_step(b)
    torch.cuda.synchronize()
    assert int((a["done"] == 1).sum()) == nc and int((b["done"] == 3).sum()) == nc
    for k in ("Hcur", "Hnew", "Unew"):
        assert np.array_equal(a[k].cpu().numpy(), b["res_val"][:, {"Hcur": 1, "Hnew": 2, "Unew": 3}[k]].cpu().numpy()), k
    assert np.array_equal(a["x"].cpu().numpy(), b["x"].cpu().numpy())
    # the host restarts a by hand exactly as the device restarted b; b's step size and length arrive one call late
    a["p"].copy_(tt(p1)); a["rem"].fill_(L1); a["dt"].copy_(tt(dt1)); a["fresh"].fill_(1)
    m.flow_step(a); m.flow_step(b)              # start evaluation of the second trajectory (b: kick deferred)
    b["dt"].copy_(tt(dt1)); b["rem"].fill_(L1)
    for s in range(L1):
        m.flow_step(a); m.flow_step(b)
    torch.cuda.synchronize()
    for k in ("x", "p", "Hcur", "Hnew", "Unew", "Ucur", "dsyn_new", "ok", "rem"):
        assert np.array_equal(a[k].cpu().numpy(), b[k].cpu().numpy()), k
    assert int(b["kick"].sum()) == 0 and int((b["done"] == 1).sum()) == nc


def test_b1_rf_entry_tiles_any_number_of_chains():
    """librf.forward / kernel_all with more chains than one launch of the RF sweeps holds (one grid row per chain): the
    host-pointer entry works through them in tiles; a chain's trace and kernels do not depend on the batch it came in."""
    from rfsurfhmc_amd.model.lib import librf
    rng = np.random.default_rng(12)
    n, nt, nchain = 3, 16, 70000
    vs = np.sort(2.5 + 1.5 * rng.random((nchain, n)), axis=1); vp = 1.75 * vs; rho = 2.2 + 0.2 * vs
    thk = 2.0 + 3.0 * rng.random((nchain, n)); thk[:, -1] = 0.0
    q = np.full((nchain, n), 9999.0)
    args = (0.045, nt, 0.5, 1.5, 2.0, "freq", 0.001, "P")
    rf = librf.forward(thk, rho, vp, vs, q, q, *args)
    assert rf.shape == (nchain, nt) and np.isfinite(rf).all()
    sub = np.r_[0:50, 32760:32790, 69950:70000]                       # chains either side of the tile boundaries
    rf2, kl2 = librf.kernel_all(thk[sub], rho[sub], vp[sub], vs[sub], q[sub], q[sub], *args)
    assert np.array_equal(rf[sub], rf2)
    rf3, kl3 = librf.kernel_all(thk[:40000], rho[:40000], vp[:40000], vs[:40000], q[:40000], q[:40000], *args)
    assert np.array_equal(rf3, rf[:40000]) and np.array_equal(kl3[sub[:80]], kl2[:80])


def test_root_that_equals_a_layer_velocity_gives_the_references_nan(orc, golden):
    """A float32 phase velocity that EQUALS a layer's float32 S velocity (about one (period, chain) item per bench step): the
    reference's sregn96 divides by that layer's vertical wavenumber and returns NaN kernels for the period (fixture written by
    the compiled reference); its samplers then end the trajectory (hmc.py:177-179).  B1 returns the same NaN pattern, the
    joint plugin a NaN gradient with flag True -- and one float32 step away everything is finite again."""
    from rfsurfhmc_amd.model.lib import libsurf
    from rfsurfhmc_amd.model.model_surf import SurfWD
    g = golden["exact_equality_reference"]
    x, t = g["x"], g["t"]
    n = len(x) // 2
    vs, thk = x[:n], x[n:]
    vp, rho, _, _ = orc.empirical_relation(vs)
    c, ka, kb, kr, kh, flag = libsurf.adjoint_kernel(thk, vp, vs, rho, t, "Rc")
    assert flag and np.array_equal(c, g["c"])
    row = int(g["nan_row"][0])
    for mine, ref in ((ka, g["ka"]), (kb, g["kb"]), (kr, g["kr"]), (kh, g["kh"])):
        assert np.array_equal(np.isnan(mine), np.isnan(ref)), (np.argwhere(np.isnan(mine) != np.isnan(ref))[:5])
        ok = np.isfinite(ref)
        assert np.abs(mine[ok] - ref[ok]).max() <= 2e-6 * np.abs(ref[ok]).max()
    m = SurfWD(tRc=t)
    d, f = m.forward(x)
    m.set_obsdata(d * 1.01)
    mis, grad, dsyn, fl = m.misfit_and_grad(x)
    assert fl and np.isfinite(mis) and np.isfinite(dsyn).all() and np.isnan(grad[:n]).all()
    # group velocities: the three passes of sregnpu; the period's U and kernels are NaN, the others finite
    u, ua, ub, ur, uh, fu = libsurf.adjoint_kernel(thk, vp, vs, rho, t[row:row + 1], "Rg")
    uo, uao, ubo, uro, uho, fo = orc.libsurf.adjoint_kernel(thk, vp, vs, rho, t[row:row + 1], "Rg")
    assert fu == fo and np.array_equal(np.isnan(u), np.isnan(uo)) and np.array_equal(np.isnan(ub), np.isnan(ubo))
    x2 = x.copy(); x2[20] = float(np.nextafter(np.float32(x[20]), np.float32(10.0)))      # the layer one float32 step faster
    vp2, rho2, _, _ = orc.empirical_relation(x2[:n])
    c2, ka2, kb2, kr2, kh2, f2 = libsurf.adjoint_kernel(x2[n:], vp2, x2[:n], rho2, t, "Rc")
    co, kao, kbo, kro, kho, fo2 = orc.libsurf.adjoint_kernel(x2[n:], vp2, x2[:n], rho2, t, "Rc")
    assert f2 and fo2 and np.isfinite(kb2).all() == np.isfinite(kbo).all()
    if np.isfinite(kbo).all():
        assert np.abs(kb2 - kbo).max() <= 2e-6 * np.abs(kbo).max()


def test_a_chain_whose_momentum_has_blown_up_is_folded_back_inside_its_bounds():
    """hmc.py:121-137 reflects a point until it is inside its bounds.  The device makes 64 reflections the reference's way and
    folds whatever is still outside in closed form (flow_mirror): a momentum of 1e6 used to leave a model with vs = 6e4 km/s,
    whose reference-semantics search scans 1.3e7 cells -- 31 s in which every other chain of the batch waited (one of 56
    chains of configs[0]'s sampler at dt 0.1, round 6).  Here: momenta of 1e5 .. 1e12 and a non-finite one, one leapfrog step:
    the end model is inside the bounds, equals the closed form, the other chains are untouched, and it takes milliseconds."""
    import time
    import torch
    import bench
    from rfsurfhmc_amd.model.model_surf import SurfWD
    thk = np.array([3., 3, 4, 5, 5, 6, 7, 8, 10, 0]); vs = np.linspace(2.9, 4.6, 10)
    tt = np.arange(5., 41.)
    x0 = np.hstack((vs, thk))
    m = SurfWD(tRc=tt, tRg=tt, device=0)
    d0, flag = m.forward(x0); assert flag
    m.set_obsdata(d0 * 1.01)
    b0 = bench.bounds_of(x0)
    dev = torch.device("cuda")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    nch = 6
    rng = np.random.default_rng(5)
    x = np.tile(x0, (nch, 1)) * (1 + 0.01 * rng.standard_normal((nch, 20))); x[:, 19] = 0.0
    p = 0.3 * rng.standard_normal((nch, 20))
    pbig = p.copy()
    pbig[1] *= 1e5 / 0.3; pbig[2] *= 1e12 / 0.3; pbig[3, 4] = np.inf; pbig[4, 2] = np.nan
    dt = np.full(nch, 0.1); L = np.ones(nch, dtype=np.int32)
    base = m.leapfrog_device(t(x), t(p), t(dt), t(L), t(b0))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = m.leapfrog_device(t(x), t(pbig), t(dt), t(L), t(b0))
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    assert el < 2.0, el
    xn = out["xnew"].cpu().numpy()
    lo, hi = b0[:, 0], b0[:, 1]
    w = hi - lo
    # (the half kick of the start leaves p - dt grad / 2: the gradient is O(1), the fold's period a few km/s -- compare chains 1, 2
    # through the closed form of their own drift, recomputed from the base run's momentum change)
    for ch in (0, 5):
        assert np.array_equal(xn[ch], base["xnew"].cpu().numpy()[ch]), ch
    for ch in (1, 2, 3, 4):
        ok = w > 0
        assert np.all(xn[ch][ok] >= lo[ok]) and np.all(xn[ch][ok] <= hi[ok]), (ch, xn[ch])
        assert np.isfinite(xn[ch]).all()
    # a chain with a non-finite momentum is rejected by its energy
    Hn = out["Hnew"].cpu().numpy()
    assert not np.isfinite(Hn[3]) or Hn[3] > 1e20
    assert not np.isfinite(Hn[4]) or Hn[4] > 1e20


def test_a_model_that_is_no_model_fails_at_once():
    """A model whose fastest layer lies thousands of km/s above the search's start value (a caller's mistake, a position that
    left its bounds) would have the reference's scan walk millions of cells of 0.005 km/s -- half a minute on the device, during
    which the other chains of the batch wait.  RootSearchT::begin fails such a search at its first evaluation (SWD_MAX_SCAN):
    the chain gets the failure return, the others are not touched."""
    import time
    import torch
    from rfsurfhmc_amd.model.model_surf import SurfWD
    thk = np.array([3., 3, 4, 5, 5, 6, 7, 8, 10, 0]); vs = np.linspace(2.9, 4.6, 10)
    tt = np.arange(5., 41.)
    x0 = np.hstack((vs, thk))
    m = SurfWD(tRc=tt, tRg=tt, device=0)
    d0, flag = m.forward(x0); assert flag
    m.set_obsdata(d0 * 1.01)
    xs = np.tile(x0, (4, 1))
    xs[1, :10] *= 2.0e4                     # vs of 6e4 .. 9e4 km/s
    xs[2, 3] = np.nan
    dev = torch.device("cuda")
    base = m.misfit_and_grad_device(torch.from_numpy(np.tile(x0, (4, 1))).to(dev))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mis, g, d, f = m.misfit_and_grad_device(torch.from_numpy(xs).to(dev))
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    assert el < 2.0, el
    f = f.cpu().numpy()
    assert f[0] == 1 and f[3] == 1 and f[1] == 0 and f[2] == 0, f
    assert float(mis[1]) == 0.0 and float(mis[2]) == 0.0 and not g[1].any() and not g[2].any()       # model_surf.py's failure return
    for i in (0, 3):
        assert torch.equal(g[i], base[1][i]) and torch.equal(mis[i], base[0][i])
