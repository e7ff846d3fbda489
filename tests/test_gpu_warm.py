"""-m gpu: the root search of the leapfrog loop inside a trajectory (include/rfsurf.h options "swd_warm_start",
"swd_warm_exact").

Inside a trajectory every (period, chain) item continues the previous step's root with the previous step's Frechet kernels
as predictor (k_swd_warm), instead of repeating the reference's sequential scan (surfdisp96.f:257-316); behind it the
reference's own refinement runs inside the reference's scan cell, groups of periods per lane (k_swd_exact, round 4), and
turns the converged roots into the reference's roots.  What is checked here:
  * every root of every step of 20-step trajectories of the 8192 bench chains against the C restatement of surfdisp96
    (bit-exact against the compiled reference): same flags; default mode: >= 99.99 % of the 6.9 M roots bit-identical, the
    rest within 1.2e-6 c, misfit and gradient of every chain and step within 1e-6 of the history-free evaluation and -- 512
    chains of the last step -- within 2e-6 / 1e-5 of the oracle's joint plugin; "swd_warm_exact" = 0 (converged roots):
    every root within 1.2e-6 c -- the reference's own refinement tolerance (surfdisp96.f:627) plus the float32 rounding
    of both values -- misfit <= 5e-5, gradient p99 <= 5e-5 (a handful of ill-conditioned chains move by tens of per cent:
    asserted as a count);
  * option 0 restores the history-free search bit for bit; models the search fails on get the reference's flags;
  * Love / group-velocity / spherical blocks; the sampler traces of the reference at the tolerance the warm start keeps;
  * "swd_exact_final": the end model of a trajectory carries reference-exact roots.
"""
import numpy as np
import pytest

# (the longest tests of the suite: their own limit, so that the global 600 s of pytest.ini -- whose watchdog ends the whole run --
# does not cut a healthy run on a slower box)
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1500)]


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


def _bench_joint(warm, n=30, nt=512, dt=0.1, exact=None, **swd):
    import bench
    from rfsurfhmc_amd.model.model_rf import ReceiverFunc
    from rfsurfhmc_amd.model.model_surf import SurfWD
    from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
    t = np.linspace(5, 44, bench.NPER)
    j = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(bench.RAY_P, nt, dt, bench.GAUSS, bench.TSHIFT, bench.WATER, "P", "freq"),
                     SurfWD(tRc=t, **swd))
    j.set_warm_start(warm)
    drf, dswd, flag = j.forward(bench.true_model(n))
    assert flag
    j.set_obsdata(drf, dswd)
    if exact is not None:
        j._ensure(n).set_option("swd_warm_exact", int(exact))
    return j, t


def _leapfrog_move(x, p, g, dt, lo, hi):
    """One leapfrog step with unit mass and mirror reflection (pyhmc/hmc.py:121-137,166-183), on torch tensors."""
    import torch
    # (a chain whose root equals a layer velocity comes back with the reference's NaN gradient -- tests/_refnan.py; the
    # reference's sampler ends the trajectory there, this bare loop lets the chain coast: a NaN model would hang the oracle's
    # restatement of the scan as it hangs the reference's)
    g = torch.nan_to_num(g, nan=0.0)
    p = p - dt * g
    x = x + dt * p
    for _ in range(4):
        over, under = x > hi, x < lo
        x = torch.where(over, 2 * hi - x, x); x = torch.where(under, 2 * lo - x, x)
        p = torch.where(over | under, -p, p)
    return x, p


@pytest.mark.parametrize("exact", [1, 0])
def test_every_root_of_20_step_trajectories_of_the_bench_chains(orc, exact):
    """8192 bench chains x 20 leapfrog steps x 40 periods: the evaluation of every step continues the one before
    (swd_warm_start = 2: the plugin entry, so that every step's synthetics come back), and every root of every step is
    compared with the C restatement of the reference's search at that step's model; misfit and gradient of every step
    with the history-free evaluation of the same models, those of the last step with the oracle's joint plugin.
    exact = 1: the default (k_swd_exact behind the warm start); 0: the converged roots as they are."""
    import os
    import torch
    import bench
    n, nt, nchain, nsteps, dt = 30, 512, 8192, 20, 0.002
    joint, t = _bench_joint(2, exact=exact)
    full, _ = _bench_joint(0)
    ctx = joint._ensure(n)
    dev = torch.device("cuda")
    bounds = bench.bounds_of(bench.true_model(n))
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    lo, hi = tt(bounds[:, 0]), tt(bounds[:, 1])
    x = tt(bench.make_models(nchain, 991206, n))
    p = tt(0.5 * np.random.default_rng(7).standard_normal((nchain, 2 * n)))
    xs_steps, c_steps, f_steps = [], [], []
    mrel, grel = [], []
    for s in range(nsteps + 1):
        m, g, d, f = joint.misfit_and_grad_device(x)
        m0, g0, d0, f0 = full.misfit_and_grad_device(x)
        assert torch.equal(f, f0), s
        # (the reference's NaN kernels where a root equals a layer velocity -- tests/_refnan.py -- in both evaluations or in
        # neither, unless the two roots differ by their last float32 step: at most a chain or two of the 8192)
        nan_w, nan_f = ~torch.isfinite(g).all(dim=1), ~torch.isfinite(g0).all(dim=1)
        # (converged roots -- exact = 0 -- are the sign changes themselves: where the sign change IS a layer velocity, the place
        # the secular function's formulas switch, they land on it far more often than the reference's nevill value, which
        # stops short of it)
        assert int((nan_w != nan_f).sum()) <= (2 if exact else 400) and int(nan_f.sum()) <= 8, (s, int(nan_w.sum()), int(nan_f.sum()))
        okd = (f0 != 0) & ~nan_w & ~nan_f
        mrel.append(((m[okd] - m0[okd]).abs() / m0[okd].abs()).cpu().numpy())
        grel.append(((g[okd] - g0[okd]).abs().amax(dim=1) / g0[okd].abs().amax(dim=1)).cpu().numpy())
        xs_steps.append(x.cpu().numpy()); c_steps.append(d[:, nt:].cpu().numpy()); f_steps.append(f.cpu().numpy() != 0)
        if s == nsteps:
            last = (x.cpu().numpy(), m.cpu().numpy(), g.cpu().numpy(), f.cpu().numpy() != 0)
        x, p = _leapfrog_move(x, p, g, dt, lo, hi)
    items, evals = ctx.stat("swd_warm_items"), ctx.stat("swd_warm_secular_evals")
    declined = ctx.stat("swd_warm_declined_chains")
    # the start models are evaluated by the full search; step 1 mirrors the out-of-range start thicknesses of the bench's
    # models back into the bounds (a move of order 1 km that no first-order model covers: those chains are handed back);
    # from then on nearly every chain is continued
    assert items >= 0.95 * (nsteps - 1) * nchain * 40, (items, declined)
    assert evals <= 4.6 * items, (evals, items)        # ~3 to refine + 1 for the branch test
    xevals = ctx.stat("swd_exact_secular_evals")
    assert (xevals > 0) == bool(exact) and xevals <= 21.0 * items, (xevals, items)
    mrel, grel = np.concatenate(mrel), np.concatenate(grel)
    from _oracle_pool import roots_batch, roots_pool, joint_batch
    worst, nident, ntot = 0.0, 0, 0
    with roots_pool() as pool:
        for s in range(nsteps + 1):
            co, oko = roots_batch(xs_steps[s], t, n, pool)
            assert np.array_equal(oko, f_steps[s]), (s, int((oko != f_steps[s]).sum()))
            ok = oko
            r = np.abs(c_steps[s][ok] - co[ok]) / co[ok]
            worst = max(worst, float(r.max()))
            nident += int((c_steps[s][ok] == co[ok]).sum()); ntot += int(ok.sum()) * 40
            # converged roots lie within 1e-6 c above the reference's (+ two float32 roundings); a root of the reference-root
            # stage that is NOT the reference's (a decision of nevill that flipped under a run-up origin 1e-9 c off) is still
            # the end of a bracket of 1e-6 c around the sign change, like the reference's: up to 2e-6 c apart
            assert r.max() <= (2.2e-6 if exact else 1.2e-6), (s, float(r.max()))
    # 512 chains of the last step against the oracle's joint plugin (the contract: 1e-5)
    xl, ml, gl, fl = last
    pick = np.nonzero(fl & np.isfinite(gl).all(axis=1))[0][:512]
    rfpar = (bench.RAY_P, nt, 0.1, bench.GAUSS, bench.TSHIFT, bench.WATER, "P", "freq")
    res = joint_batch(xl[pick], rfpar, t, joint.dobs[:nt], joint.dobs[nt:])
    om = np.array([r[0] for r in res]); og = np.array([r[1] for r in res]); of = np.array([r[3] for r in res])
    assert of.all()
    omr = np.abs(ml[pick] - om) / np.abs(om)
    ogr = np.abs(gl[pick] - og).max(axis=1) / np.abs(og).max(axis=1)
    print(f"exact = {exact}: {ntot} roots over {nsteps + 1} steps, worst {worst:.3e} c, bit-identical {nident} ({nident / ntot:.5%}); "
          f"{evals / max(items, 1):.2f} + {xevals / max(items, 1):.2f} secular evaluations per item, {declined} chain evaluations handed back; "
          f"vs the history-free evaluation: misfit max {mrel.max():.2e}, gradient max {grel.max():.2e} p99 {np.quantile(grel, 0.99):.2e}; "
          f"vs the oracle (512 chains): misfit max {omr.max():.2e}, gradient max {ogr.max():.2e} p99 {np.quantile(ogr, 0.99):.2e}")
    if exact:
        assert nident >= 0.9999 * ntot, (nident, ntot)
        # (round 5's default -- one run-up period, origins to 5e-7 c -- leaves 5e-5 of the roots one float32 step off the
        # sequential search's: measured 99.9954 % identical, misfit within 1.9e-7, gradient within 2.0e-6 of it; round 4's two
        # run-up periods: 99.998 %, 5e-8, 6e-8)
        assert mrel.max() <= 1e-6 and grel.max() <= 5e-6, (mrel.max(), grel.max())
        assert omr.max() <= 2e-6 and ogr.max() <= 1e-5, (omr.max(), ogr.max())
    else:
        # converged roots, 0.5 .. 1e-6 c above the reference's: a misfit built on residuals of ~0.3 km/s moves by ~1.4e-5;
        # gradients by ~1e-5, except on the ill-conditioned chains (fewer than 1 %)
        assert mrel.max() <= 5e-5 and np.quantile(grel, 0.99) <= 5e-5, (mrel.max(), np.quantile(grel, 0.99))
        assert (grel > 1e-3).mean() <= 0.01
        assert omr.max() <= 5e-5 and np.quantile(ogr, 0.99) <= 5e-5, (omr.max(), np.quantile(ogr, 0.99))


def test_roots_under_large_steps_against_the_restatement(orc):
    """The same check at step sizes of a dual-averaging run (dt = 0.03 and 0.08: first-order changes of the roots of
    several 1e-2 km/s, beyond the point where the first-order model alone brackets the root): 1024 chains x 8 steps;
    whatever the continuation accepts is within 1.2e-6 c of the reference search, whatever it cannot follow is handed back."""
    import os
    import torch
    import bench
    n, nt, nchain, nsteps = 30, 512, 1024, 8
    dev = torch.device("cuda")
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    bounds = bench.bounds_of(bench.true_model(n))
    lo, hi = tt(bounds[:, 0]), tt(bounds[:, 1])
    from _oracle_pool import roots_batch, roots_pool
    with roots_pool() as pool:
        for dt in (0.03, 0.08):
            joint, t = _bench_joint(2)
            ctx = joint._ensure(n)
            xs = np.clip(bench.make_models(nchain, 17, n), bounds[:, 0], bounds[:, 1])
            x = tt(xs); p = tt(0.5 * np.random.default_rng(3).standard_normal(xs.shape))
            worst = 0.0
            for s in range(nsteps + 1):
                m, g, d, f = joint.misfit_and_grad_device(x)
                co, oko = roots_batch(x.cpu().numpy(), t, n, pool)
                fl = f.cpu().numpy() != 0
                assert np.array_equal(oko, fl), (dt, s)
                r = np.abs(d[:, nt:].cpu().numpy()[oko] - co[oko]) / co[oko]
                assert r.max() <= 1.2e-6, (dt, s, float(r.max()))
                worst = max(worst, float(r.max()))
                # (no gradient in the move: the random momentum alone sets the step length, as intended here)
                x, p = _leapfrog_move(x, p, torch.zeros_like(g), dt, lo, hi)
            items, decl = ctx.stat("swd_warm_items"), ctx.stat("swd_warm_declined_chains")
            print(f"dt {dt}: worst {worst:.3e} c; {items / (nsteps * nchain * 40):.2%} of the items continued, "
                  f"{decl} chain evaluations handed back, {ctx.stat('swd_warm_secular_evals') / max(items, 1):.2f} evaluations per item")
            assert items >= 0.5 * nsteps * nchain * 40


def test_option_zero_is_the_history_free_search_and_failing_models_keep_their_flags(golden):
    """swd_warm_start = 0 inside the flow entry == the plugin evaluation of the same models, bit for bit; with the warm
    start on, models the reference search fails on (unsorted velocities with strong inversions) get the same flags and the
    same failure returns, step after step."""
    import torch
    import bench
    n, nt, nchain = 30, 512, 1024
    dev = torch.device("cuda")
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rng = np.random.default_rng(11)
    xs = bench.make_models(nchain, 5, n)
    wild = rng.random(nchain) < 0.3                         # a third of the chains: unsorted layers
    for i in np.nonzero(wild)[0]:
        xs[i, :n] = rng.permutation(xs[i, :n])
    bounds = np.stack([np.r_[np.full(n, 1.5), np.full(n, 0.0)], np.r_[np.full(n, 5.0), np.full(n, 3.0)]], axis=1)
    lo, hi = tt(bounds[:, 0]), tt(bounds[:, 1])
    jw, _ = _bench_joint(2); je, _ = _bench_joint(0)
    x = tt(xs); p = tt(0.5 * rng.standard_normal(xs.shape))
    nfail = nroot = nsame = 0
    for s in range(8):
        mw, gw, dw, fw = jw.misfit_and_grad_device(x)
        me, ge, de, fe = je.misfit_and_grad_device(x)
        assert torch.equal(fw, fe), s
        bad = fe == 0
        nfail += int(bad.sum())
        assert torch.equal(mw[bad], me[bad]) and torch.equal(gw[bad], ge[bad]) and torch.equal(dw[bad], de[bad])
        ok = ~bad
        r = ((dw[ok][:, nt:] - de[ok][:, nt:]).abs() / de[ok][:, nt:]).max().item()
        # (the reference-root stage: identical roots but for a flipped decision of nevill now and then -- both values are
        # then ends of a 1e-6 c bracket around the same sign change: 2e-6 c apart at most)
        assert r <= 2.2e-6, (s, r)
        nroot += int(ok.sum()) * 40; nsame += int((dw[ok][:, nt:] == de[ok][:, nt:]).sum())
        x, p = _leapfrog_move(x, p, torch.where(bad[:, None], torch.zeros_like(gw), gw), 0.002, lo, hi)
    # (unsorted models have crowded spectra: the secular function -- normalised to a largest vector component of 1 -- is a step
    # at the root and saturated elsewhere, and nevill's decisions there (is the midpoint's value between the ends'?  :630-634)
    # compare numbers that differ in their last bits: the reference's own root depends on the rounding of its arithmetic.  A
    # sequential search and a period-parallel one, or two builds of the reference, part ways on ~1 % of such roots)
    print(f"unsorted third of the chains: {nsame} of {nroot} roots identical to the history-free search")
    assert nsame >= 0.995 * nroot, (nsame, nroot)
    # models the reference's search FAILS on (tests/golden/swd_reference.npz "inverted_20": ierr = 1 in the compiled
    # reference) beside models it solves, moving together: same flags and failure returns at every step, and the failing
    # chains never poison their neighbours
    from rfsurfhmc_amd.model.model_surf import SurfWD
    gs = golden["swd_reference"]
    thk, vs, tp = gs["inverted_20/thk"], gs["inverted_20/vs"], gs["inverted_20/t"]
    nl = len(vs)
    x_bad = np.hstack((vs, thk)); x_ok = np.hstack((np.linspace(2.5, 4.2, nl), np.full(nl, 3.0)))
    xs2 = np.vstack([x_bad if i % 3 == 0 else x_ok for i in range(192)]) * (1 + 0.002 * rng.standard_normal((192, 2 * nl)))
    sw, se = SurfWD(tRc=tp), SurfWD(tRc=tp)
    sw.set_warm_start(2); se.set_warm_start(0)
    dobs = np.full(len(tp), 3.0)
    sw.set_obsdata(dobs); se.set_obsdata(dobs)
    x2 = tt(xs2); p2 = tt(0.5 * rng.standard_normal(xs2.shape))
    lo2, hi2 = tt(0.5 * xs2.min(0)), tt(1.5 * xs2.max(0))
    for s in range(6):
        mw, gw, dw, fw = sw.misfit_and_grad_device(x2)
        me, ge, de, fe = se.misfit_and_grad_device(x2)
        assert torch.equal(fw, fe), s
        bad = fe == 0
        nfail += int(bad.sum())
        assert torch.equal(mw[bad], me[bad]) and torch.equal(gw[bad], ge[bad]) and torch.equal(dw[bad], de[bad])
        assert ((dw[~bad] - de[~bad]).abs() / de[~bad]).max().item() <= 1.2e-6
        x2, p2 = _leapfrog_move(x2, p2, gw, 0.002, lo2, hi2)
    assert nfail >= 6 * 60                                    # the inverted models do fail, at every step
    # flow entry with the option off: every step's numbers are those of the plain evaluation
    j0, _ = _bench_joint(0)
    st = j0.flow_state(tt(xs[:256]), torch.full((256,), 0.002, dtype=torch.float64, device=dev), tt(bounds))
    st["p"].copy_(tt(0.5 * rng.standard_normal((256, 2 * n)))); st["rem"].fill_(3); st["fresh"].fill_(1)
    for _ in range(4):
        j0.flow_step(st)
    torch.cuda.synchronize()
    done = st["done"].cpu().numpy() == 1
    m, g, d, f = je.misfit_and_grad_device(st["x"].clone())
    okc = torch.from_numpy(done).to(dev) & (f != 0) & (st["ok"] != 0)
    assert int(okc.sum()) > 100
    assert torch.equal(st["Unew"][okc], m[okc]) and torch.equal(st["dsyn_new"][okc], d[okc])


def test_dense_grid_walk_gives_the_speculative_walk_s_verdicts():
    """Option swd_walk_dense: the later periods of walking sequences evaluate exactly the grid points up to the continued
    root, densely packed -- the same verdicts as rounds of 8 speculative lanes per item: two plugins on the same models
    (2048 chains, half of them unsorted, 10 steps at a sampler's step size) hand back the same number of chains for the
    same causes at every step and return identical numbers, with fewer evaluations."""
    from _refnan import same
    import torch
    import bench
    n, nt, nchain = 30, 512, 2048
    dev = torch.device("cuda")
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rng = np.random.default_rng(23)
    bounds = bench.bounds_of(bench.true_model(n))
    xs = np.clip(bench.make_models(nchain, 5, n), bounds[:, 0], bounds[:, 1])
    for i in np.nonzero(rng.random(nchain) < 0.5)[0]:
        xs[i, :n] = rng.permutation(xs[i, :n])
    lo, hi = tt(bounds[:, 0]), tt(bounds[:, 1])
    jd, _ = _bench_joint(2); js, _ = _bench_joint(2)
    cd, cs = jd._ensure(n), js._ensure(n)
    cd.set_option("swd_walk_dense", 1); cs.set_option("swd_walk_dense", 0)
    cd.set_option("swd_walk_window", -1); cs.set_option("swd_walk_window", -1)      # (every period of a walking sequence: the comparison's subject)
    x = tt(xs); p = tt(0.5 * rng.standard_normal(xs.shape))
    names = ("swd_warm_declined_chains", "swd_warm_walked_chains", "swd_exact_declined_chains", "swd_warm_items")
    for s in range(11):
        md, gd, dd, fd = jd.misfit_and_grad_device(x)
        ms, gs, ds, fs = js.misfit_and_grad_device(x)
        assert torch.equal(fd, fs) and torch.equal(md, ms) and same(gd, gs) and torch.equal(dd, ds), s      # (NaN == NaN: tests/_refnan.py)
        assert [cd.stat(k) for k in names] == [cs.stat(k) for k in names], (s, [cd.stat(k) for k in names], [cs.stat(k) for k in names])
        x, p = _leapfrog_move(x, p, torch.zeros_like(gd), 0.03, lo, hi)
    walked, items = cd.stat("swd_warm_walked_chains"), cd.stat("swd_warm_items")
    ed, es = cd.stat("swd_warm_secular_evals"), cs.stat("swd_warm_secular_evals")
    print(f"{walked} chain evaluations walked the grid, {cd.stat('swd_warm_declined_chains')} handed back; warm-start + branch-test "
          f"evaluations per item: dense {ed / items:.2f}, speculative {es / items:.2f}")
    assert walked >= 2000 and ed < 0.85 * es


def test_exact_final_gives_reference_roots_at_the_end_model():
    """swd_exact_final = 1: the steps inside a trajectory are warm-started, the evaluation of its END model goes through
    the reference-semantics search: the synthetics stored for that model are bit for bit those of a plain evaluation."""
    import torch
    import bench
    n, nt, nchain, L = 30, 512, 512, 6
    dev = torch.device("cuda")
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    jw, _ = _bench_joint(1); je, _ = _bench_joint(0)
    ctx = jw._ensure(n)
    ctx.set_option("swd_exact_final", 1)
    bounds = bench.bounds_of(bench.true_model(n))
    xs = np.clip(bench.make_models(nchain, 3, n), bounds[:, 0], bounds[:, 1])
    st = jw.flow_state(tt(xs), torch.full((nchain,), 0.002, dtype=torch.float64, device=dev), tt(bounds))
    st["p"].copy_(tt(0.5 * np.random.default_rng(2).standard_normal(xs.shape))); st["rem"].fill_(L); st["fresh"].fill_(1)
    for _ in range(L + 1):
        jw.flow_step(st)
    torch.cuda.synchronize()
    assert int((st["done"] == 1).sum()) == nchain
    items = ctx.stat("swd_warm_items")
    assert items >= 0.9 * (L - 1) * nchain * 40                     # the inner steps were warm-started
    m, g, d, f = je.misfit_and_grad_device(st["x"].clone())
    okc = st["ok"] != 0               # (a trajectory that met the reference's NaN gradient on its way has ended there: tests/_refnan.py)
    assert int((~okc).sum()) <= 2
    assert torch.equal(st["dsyn_new"][okc], d[okc]) and torch.equal(st["Unew"][okc], m[okc])


def test_love_group_and_sphere_blocks_are_continued_too(orc):
    """All four blocks (Rc, Rg, Lc, Lg: six search sequences -- the phase periods, shared by the group blocks' central pass,
    and the +-5 % passes of the two group kernels) on a flat and
    on a flattened earth: the warm-started sequence of evaluations against the history-free one."""
    import torch
    from rfsurfhmc_amd.model.model_surf import SurfWD
    rng = np.random.default_rng(4)
    n, nchain = 12, 256
    thk = np.r_[np.full(n - 1, 3.0), 0.0]; vs = np.linspace(2.8, 4.5, n)
    x0 = np.hstack((vs, thk))
    t = np.linspace(6.0, 36.0, 7)
    dev = torch.device("cuda")
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for sph in (False, True):
        kw = dict(tRc=t, tRg=t, tLc=t, tLg=t, sphere=sph, reference_periods=False)
        sw, se = SurfWD(**kw), SurfWD(**kw)
        sw.set_warm_start(2); se.set_warm_start(0)
        d0, fl = se.forward(x0)
        assert fl
        sw.set_obsdata(d0 * 1.01); se.set_obsdata(d0 * 1.01)
        xs = np.tile(x0, (nchain, 1)) * (1 + 0.02 * rng.standard_normal((nchain, 2 * n)))
        xs[:, :n] = np.sort(xs[:, :n], axis=1); xs[:, -1] = 0.0
        x = tt(xs); p = tt(0.5 * rng.standard_normal(xs.shape))
        lo, hi = tt(0.7 * xs.min(0)), tt(1.3 * xs.max(0) + 1e-9)
        for s in range(6):
            mw, gw, dw, fw = sw.misfit_and_grad_device(x)
            me, ge, de, fe = se.misfit_and_grad_device(x)
            assert torch.equal(fw, fe) and bool((fe != 0).all())
            nt4 = len(t)
            # phase blocks: the roots themselves; group blocks: U = (k I1 + I2) / (omega I0) is evaluated AT the root and
            # moves by ~1e2 times the root's relative change (the reference's own U carries its 1e-6 c root tolerance
            # the same way): 2e-4
            for b, tol in ((0, 1.2e-6), (2, 1.2e-6), (1, 2e-4), (3, 2e-4)):
                a_, b_ = dw[:, b * nt4:(b + 1) * nt4], de[:, b * nt4:(b + 1) * nt4]
                assert ((a_ - b_).abs() / b_.abs()).max().item() <= tol, (sph, s, b)
            x, p = _leapfrog_move(x, p, gw, 0.003, lo, hi)
        ctx = sw._ensure(n)
        items = ctx.stat("swd_warm_items")
        assert items >= 0.9 * 5 * nchain * 6 * len(t), (sph, items)


@pytest.mark.parametrize("kind", ["hmc", "da"])
def test_reference_sampler_traces_with_the_warm_start_on(kind, golden):
    """The reference's own sampler traces (tests/golden/sampler_hybrid.npz) with the warm start on (default: with the
    reference-root stage behind it): same initial models, L draws and accept decisions; states, step sizes and misfits at
    the tolerances tests/test_gpu_samplers.py holds the history-free search to (1e-6 / 1e-5) -- the steps inside a
    trajectory now return the reference's own roots."""
    from test_gpu_samplers import _joint
    from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
    from rfsurfhmc_amd.pyhmc.hmcda import HMCDualAveraging
    g = golden["sampler_hybrid"]
    joint = _joint(g, warm=1)
    if kind == "hmc":
        s = HamitonianMC(joint, g["bounds"], 0.1, [5, 20], 2, 991206, 6, 3, myrank=0, name="t", outdir=None,
                         nchains=2, verbose=False)
        s.trace = []
        mis = s.sample()
        for c, tag in ((0, "hmc_r0"), (1, "hmc_r1")):
            seq = [(tr, tr["active"].index(c)) for tr in s.trace if c in tr["active"]]
            assert np.array_equal(np.array([tr["L"][k] for tr, k in seq]), g[f"{tag}/L"])
            assert np.array_equal(np.array([tr["accept"][k] for tr, k in seq]), g[f"{tag}/accept"])
            assert rel(np.array([tr["xres"][k] for tr, k in seq]), g[f"{tag}/x"]) < 1e-6
            assert rel(np.array([tr["Ures"][k] for tr, k in seq]), g[f"{tag}/U"]) < 1e-5
            assert rel(mis[c], g[f"{tag}/misfit"]) < 1e-5
    else:
        s = HMCDualAveraging(joint, g["bounds"], 0.1, 10, 2, 0.65, 991206, 6, 3, myrank=0, name="t", outdir=None,
                             nchains=1, verbose=False)
        s.trace = []
        mis = s.sample()
        assert np.array_equal(np.array([tr["L"][0] for tr in s.trace]), g["da_r0/L"])
        # (dual averaging feeds every acceptance ratio back into the next step size: differences compound)
        assert rel(np.array([tr["dt"][0] for tr in s.trace]), g["da_r0/dt"]) < 1e-6
        assert rel(np.array([tr["xend"][0] for tr in s.trace]), g["da_r0/x"]) < 1e-6
        assert rel(mis, g["da_r0/misfit"]) < 1e-5


def test_sampler_with_and_without_the_warm_start_samples_alike(golden):
    """End to end: the same seeded HamitonianMC run (256 chains, the 7-layer joint problem of the reference's traces) with
    the warm start on (reference-root stage included: the default) and off.  Every chain consumes the same random numbers
    in both and the steps inside a trajectory return the reference's roots, so every chain takes the same decisions and
    the samples agree to the few roots in 1e5 that end on a neighbouring float32 (1e-6 of a sample at most)."""
    from test_gpu_samplers import _joint
    from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
    g = golden["sampler_hybrid"]
    nc = 256
    rng = np.random.default_rng(8)
    x0 = np.clip(g["x0"][None, :] * (1 + 0.03 * rng.standard_normal((nc, len(g["x0"])))), g["bounds"][:, 0], g["bounds"][:, 1])
    runs = {}
    for warm in (0, 1):
        s = HamitonianMC(_joint(g, warm=warm), g["bounds"], 0.02, [5, 20], 2, 991206, 12, 4, myrank=0, name="t", outdir=None,
                         nchains=nc, verbose=False)
        mis = s.sample_flow(x_init=x0, max_steps=4000)
        assert s.finished
        runs[warm] = (np.asarray(mis), np.asarray(s.x_cache), np.asarray(s.naccepted))
    (m0, x0s, a0), (m1, x1s, a1) = runs[0], runs[1]
    same = a0 == a1
    assert same.all(), same.mean()
    relx = np.abs(x1s[same] - x0s[same]).max() / np.abs(x0s[same]).max()
    relm = np.abs(m1[same] - m0[same]).max() / np.abs(m0[same]).max()
    print(f"warm vs full-search sampler: {same.mean():.1%} of the chains with identical accept counts; on those samples differ "
          f"by {relx:.2e}, misfits by {relm:.2e}; ensemble mean misfit {m0[:, -1].mean():.6f} vs {m1[:, -1].mean():.6f}")
    assert relx < 1e-6 and relm < 1e-5
    assert abs(m1[:, -1].mean() - m0[:, -1].mean()) <= 0.01 * m0[:, -1].std()
    assert np.all(np.abs(x1s[:, -1].mean(0) - x0s[:, -1].mean(0)) <= 0.02 * x0s[:, -1].std(0) + 1e-12)


def test_chains_that_sit_a_step_out_sample_the_same(golden):
    """rfs_set_option flow_async_handback (the samplers' default): a chain whose root search is handed back to the
    reference-semantics search sits that device step out and its search runs beside the following steps.  The same seeded
    HamitonianMC run -- the bench's 30-layer joint problem at its step size, 512 chains from burned-in models, where a few
    chains per step are handed back -- with the option on and off: every chain goes through the same models and decisions,
    so the accept counts are equal and the samples agree to the last bits of the roots."""
    import torch
    import bench
    from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
    n, nc = 30, 512
    joint, t = _bench_joint(1)
    bounds = bench.bounds_of(bench.true_model(n))
    xs = bench.make_models(nc, 4, n)
    ctx = joint._ensure(n)
    # burn in (one run, then both variants start from its end models)
    keep = {}
    s0 = HamitonianMC(joint, bounds, 0.05, [5, 20], 2, 991206, 40, 4, myrank=0, name="t", outdir=None, nchains=nc, verbose=False)
    s0.sample_flow(x_init=xs, max_steps=161, step_hook=lambda s, st: keep.__setitem__("x", st["x"].clone()) if s == 160 else None)
    xb = keep["x"].cpu().numpy()
    import os
    MODES = (1, 0) if os.environ.get("RFS_TEST_SYNC_TWICE") != "1" else (2, 0)
    runs = {}
    for mode in MODES:
        d0 = ctx.stat("swd_warm_declined_chains")
        s = HamitonianMC(joint, bounds, 0.05, [5, 20], 2, 991206, 30, 4, myrank=0, name="t", outdir=None, nchains=nc, verbose=False)
        # (a fixed number of device steps: on these rough models a few chains sit on a discontinuity of the reference's
        # dispersion curve -- a mode jump -- and never accept, as they would not in the reference)
        mis = s.sample_flow(x_init=xb, max_steps=240, async_handback=bool(mode & 1))
        runs[mode & 1 if mode < 2 else 1] = (np.asarray(mis), np.asarray(s.x_cache), np.asarray(s.naccepted), np.asarray(s.ntrajectories),
                      ctx.stat("swd_warm_declined_chains") - d0)
    (m1, x1, a1, n1, h1), (m0, x0, a0, n0, h0) = runs[1], runs[0]
    assert h1 > 0 and h0 > 0, (h1, h0)                  # chains were handed back in both
    # a chain that sat steps out is a few leapfrog steps behind at the cut: compare the samples both runs have
    k = np.minimum((m1 != 0).sum(axis=1), (m0 != 0).sum(axis=1))
    # (how far behind is a matter of timing -- a chain on a stretch where the reference's search runs in its retry mode is handed
    # back at every step and advances one leapfrog step per search, a few device steps each; a chain whose evaluations fail ends
    # a trajectory per completed search -- never ahead)
    assert (n1 <= n0).all() and (k > 3).mean() > 0.8, (int((n0 - n1).max()), float((k > 3).mean()))
    relx = relm = 0.0
    for c in range(nc):
        if k[c] > 0:
            relx = max(relx, np.abs(x1[c, :k[c]] - x0[c, :k[c]]).max() / np.abs(x0[c, :k[c]]).max())
            relm = max(relm, np.abs(m1[c, :k[c]] - m0[c, :k[c]]).max() / np.abs(m0[c, :k[c]]).max())
    print(f"sitting out vs waiting: {h1} / {h0} chain evaluations handed back; {int(k.sum())} samples compared, "
          f"models differ by {relx:.2e}, misfits by {relm:.2e}")
    assert relx <= 1e-6 and relm <= 1e-5


def test_wide_brackets_and_skipped_idle_chains_sample_the_same():
    """Round 5: a warm search that finds no sign change within its trust radius keeps widening (the grid walk vouches for
    what it finds, "swd_warm_widen"), and chains that are idle in a flow step are neither continued nor handed back
    ("flow_skip_idle").  Both change only WHO evaluates a chain -- the period-parallel stages or the sequential search --
    never the roots: the same seeded sampler run from burned-in models with both options on and both off (round 4's
    behaviour; blocking hand-backs, so that device steps line up) accepts the same trajectories and stores the same
    samples; fewer chains go to the full search with them on."""
    import bench
    from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
    n, nc = 30, 1024
    joint, t = _bench_joint(1)
    bounds = bench.bounds_of(bench.true_model(n))
    ctx = joint._ensure(n)
    keep = {}
    s0 = HamitonianMC(joint, bounds, 0.05, [5, 20], 2, 991206, 40, 4, myrank=0, name="t", outdir=None, nchains=nc, verbose=False)
    s0.sample_flow(x_init=bench.make_models(nc, 4, n), max_steps=161,
                   step_hook=lambda s, st: keep.__setitem__("x", st["x"].clone()) if s == 160 else None)
    xb = keep["x"].cpu().numpy()
    runs = {}
    for on in (1, 0):
        ctx.set_option("swd_warm_widen", on); ctx.set_option("flow_skip_idle", on)
        d0 = ctx.stat("swd_warm_declined_chains"); w0 = ctx.stat("swd_warm_wide_chains")
        s = HamitonianMC(joint, bounds, 0.05, [5, 20], 2, 991206, 30, 4, myrank=0, name="t", outdir=None, nchains=nc, verbose=False)
        mis = s.sample_flow(x_init=xb, max_steps=160, async_handback=False)
        runs[on] = (np.asarray(mis), np.asarray(s.x_cache), np.asarray(s.naccepted), np.asarray(s.ntrajectories),
                    ctx.stat("swd_warm_declined_chains") - d0, ctx.stat("swd_warm_wide_chains") - w0)
    ctx.set_option("swd_warm_widen", 1); ctx.set_option("flow_skip_idle", 1)
    (m1, x1, a1, n1, h1, w1), (m0, x0, a0, n0, h0, w0_) = runs[1], runs[0]
    assert np.array_equal(n1, n0)                        # blocking hand-backs: every chain completed the same trajectories
    same = a1 == a0
    k = np.minimum((m1 != 0).sum(axis=1), (m0 != 0).sum(axis=1))
    relx = relm = 0.0
    for c in np.nonzero(same)[0]:
        if k[c] > 0:
            relx = max(relx, np.abs(x1[c, :k[c]] - x0[c, :k[c]]).max() / np.abs(x0[c, :k[c]]).max())
            relm = max(relm, np.abs(m1[c, :k[c]] - m0[c, :k[c]]).max() / np.abs(m0[c, :k[c]]).max())
    print(f"options on / off: {h1} / {h0} chain evaluations handed back, {w1} / {w0_} chains walked the grid for a wide bracket; "
          f"{int(same.sum())} of {nc} chains with identical accept counts, {int(k[same].sum())} samples compared: "
          f"models differ by {relx:.2e}, misfits by {relm:.2e}")
    assert h1 < h0 and w1 > w0_
    # (a root off by a float32 step can flip an acceptance whose draw lies within 1e-7 of the threshold: not more than a chain or two)
    assert same.mean() >= 0.995 and relx <= 1e-6 and relm <= 1e-5


def test_warm_search_in_rounds_is_the_one_round_search_bit_for_bit():
    """Round 5: k_swd_warm runs in rounds -- a budget of evaluations per lane, the unfinished searches packed densely for the
    next round ("swd_warm_round_budgets": a wavefront executes what its slowest lane needs, and the searches are very uneven).
    Which wavefront a search finishes in changes none of its evaluations, and neither does the last round's 16 lanes per
    search (k_swd_warm_coop: the single lane's arithmetic, shared out): the same seeded sampler run from burned-in models in
    one round, in the default three with and without the cooperative last round, and in four rounds with tiny budgets (which
    overflow their lists) gives identical samples, misfits and counters."""
    import bench
    from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
    n, nc = 30, 1024
    joint, t = _bench_joint(1)
    bounds = bench.bounds_of(bench.true_model(n))
    ctx = joint._ensure(n)
    keep = {}
    s0 = HamitonianMC(joint, bounds, 0.05, [5, 20], 2, 991206, 40, 4, myrank=0, name="t", outdir=None, nchains=nc, verbose=False)
    s0.sample_flow(x_init=bench.make_models(nc, 4, n), max_steps=121,
                   step_hook=lambda s, st: keep.__setitem__("x", st["x"].clone()) if s == 120 else None)
    xb = keep["x"].cpu().numpy()
    runs = {}
    try:
        for budgets, coop in ((0, 1), (302, 1), (302, 0), (10101, 1), (303, 1), (4, 1), (5, 0)):
            ctx.set_option("swd_warm_round_budgets", budgets); ctx.set_option("swd_warm_last_round_coop", coop)
            names = ("swd_warm_declined_chains", "swd_warm_secular_evals", "swd_warm_items", "swd_exact_secular_evals")
            c0 = [ctx.stat(k) for k in names]
            p1 = ctx.stat("swd_warm_passed_on_1")
            s = HamitonianMC(joint, bounds, 0.05, [5, 20], 2, 991206, 30, 4, myrank=0, name="t", outdir=None, nchains=nc, verbose=False)
            mis = s.sample_flow(x_init=xb, max_steps=100, async_handback=False)
            runs[(budgets, coop)] = (np.asarray(mis), np.asarray(s.x_cache), np.asarray(s.naccepted), np.asarray(s.ntrajectories),
                             [ctx.stat(k) - v for k, v in zip(names, c0)], ctx.stat("swd_warm_passed_on_1") - p1)
    finally:
        ctx.set_option("swd_warm_round_budgets", 303); ctx.set_option("swd_warm_last_round_coop", 1)
    ref = runs[(0, 1)]
    # (the rounds did take place; budgets of one evaluation overflow the lists -- a quarter of the items -- so their count is capped)
    assert ref[5] == 0 and runs[(302, 1)][5] > 0 and runs[(10101, 1)][5] > 0
    for budgets in ((302, 1), (302, 0), (10101, 1), (303, 1), (4, 1), (5, 0)):
        r = runs[budgets]
        assert np.array_equal(r[0], ref[0]) and np.array_equal(r[1], ref[1]), budgets
        assert np.array_equal(r[2], ref[2]) and np.array_equal(r[3], ref[3]), budgets
        # (hand-backs, items and the stage's evaluations identical; the warm stage's count includes the branch test's, which skips an
        # item whose chain a sister item has already handed back -- which of the two runs first is a matter of scheduling)
        assert r[4][0] == ref[4][0] and r[4][2] == ref[4][2] and abs(r[4][1] - ref[4][1]) <= 1e-5 * ref[4][1], (budgets, r[4], ref[4])
        assert abs(r[4][3] - ref[4][3]) <= 1e-3 * ref[4][3], (budgets, r[4], ref[4])


def test_reference_root_stage_with_16_lanes_per_group_is_the_single_lane_stage_bit_for_bit():
    """Round 6: k_swd_exact_coop (16 lanes per group of periods: every lane builds the entries of every 16th layer, all run the
    short recurrence) against k_swd_exact (a lane per group) -- the same seeded sampler run from burned-in models, with the
    form forced on ("swd_exact_coop" 2) and off (0), stores identical samples, misfits and counters (joint problem of the
    bench, 512 chains; and the SWD-only problem of configs[0] with group velocities, 16 chains)."""
    import bench
    from rfsurfhmc_amd.model.model_surf import SurfWD
    from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
    n, nc = 30, 512
    joint, t = _bench_joint(1)
    bounds = bench.bounds_of(bench.true_model(n))
    ctx = joint._ensure(n)
    keep = {}
    s0 = HamitonianMC(joint, bounds, 0.05, [5, 20], 2, 991206, 40, 4, myrank=0, name="t", outdir=None, nchains=nc, verbose=False)
    s0.sample_flow(x_init=bench.make_models(nc, 4, n), max_steps=121,
                   step_hook=lambda s, st: keep.__setitem__("x", st["x"].clone()) if s == 120 else None)
    xb = keep["x"].cpu().numpy()

    def runs_of(model, ctx, bounds, x_init, nchains, dt):
        out = {}
        try:
            ctx.set_option("swd_exact_redo_runup", 0)       # (the second try of small batches is the 16-lane form's alone: off for the comparison)
            ctx.set_option("swd_exact_group_small", 4)      # (... and so are their shorter groups: the same groups in both forms)
            for coop in (0, 2):
                ctx.set_option("swd_exact_coop", coop)
                names = ("swd_warm_declined_chains", "swd_exact_declined_chains", "swd_exact_secular_evals", "swd_warm_items")
                c0 = [ctx.stat(k) for k in names]
                s = HamitonianMC(model, bounds, dt, [5, 20], 2, 991206, 30, 4, myrank=0, name="t", outdir=None, nchains=nchains, verbose=False)
                mis = s.sample_flow(x_init=x_init, max_steps=80, async_handback=False)
                out[coop] = (np.asarray(mis), np.asarray(s.x_cache), np.asarray(s.naccepted), np.asarray(s.ntrajectories),
                             [ctx.stat(k) - v for k, v in zip(names, c0)])
        finally:
            ctx.set_option("swd_exact_coop", 1); ctx.set_option("swd_exact_redo_runup", -1); ctx.set_option("swd_exact_group_small", 2)
        return out

    r = runs_of(joint, ctx, bounds, xb, nc, 0.05)
    assert r[0][4][2] > 0                                                # the stage ran
    for i in range(4):
        assert np.array_equal(r[0][i], r[2][i]), i
    _same_counters(r[0][4], r[2][4])
    # ... and in ROUNDS ("swd_exact_budget"): a lane per group with a budget of evaluations, the unfinished groups' machines
    # saved in the middle of whatever they were doing and continued by the 16-lane kernel -- budgets that pass on a few per
    # cent of the groups, most of them, and (5) every one, the last also overflowing the list
    try:
        ctx.set_option("swd_exact_coop", 0)
        rb = {}
        for budget in (0, 44, 30, 5):
            ctx.set_option("swd_exact_budget", budget)
            s = HamitonianMC(joint, bounds, 0.05, [5, 20], 2, 991206, 30, 4, myrank=0, name="t", outdir=None, nchains=nc, verbose=False)
            mis = s.sample_flow(x_init=xb, max_steps=80, async_handback=False)
            rb[budget] = (np.asarray(mis), np.asarray(s.x_cache), np.asarray(s.naccepted), np.asarray(s.ntrajectories))
    finally:
        ctx.set_option("swd_exact_budget", 44); ctx.set_option("swd_exact_coop", 1)
    for budget in (0, 44, 30, 5):
        for i in range(4):
            assert np.array_equal(rb[budget][i], r[0][i]), (budget, i)
    # ... and with the second launch BESIDE the eigenfunction pass of all items ("swd_exact_overlap": the background form of the
    # flow entries -- hand-backs sit steps out --, the groups it finishes get their eigenfunctions again afterwards)
    try:
        ctx.set_option("swd_exact_coop", 0); ctx.set_option("swd_exact_budget", 40)
        ro = {}
        for ov in (0, 1):
            ctx.set_option("swd_exact_overlap", ov)
            s = HamitonianMC(joint, bounds, 0.05, [5, 20], 2, 991206, 30, 4, myrank=0, name="t", outdir=None, nchains=nc, verbose=False)
            mis = s.sample_flow(x_init=xb, max_steps=80, async_handback=True)
            ro[ov] = (np.asarray(mis), np.asarray(s.x_cache), np.asarray(s.naccepted), np.asarray(s.ntrajectories))
    finally:
        ctx.set_option("swd_exact_budget", 44); ctx.set_option("swd_exact_coop", 1); ctx.set_option("swd_exact_overlap", 0)
    # (stored samples, their misfits and the accept counts: with hand-backs in the background the device step in which a chain's
    # search is found complete is a matter of timing, so the number of trajectories INSIDE a fixed number of steps may differ)
    for i in range(3):
        assert np.array_equal(ro[0][i], ro[1][i]), i
    # configs[0]'s plugin: 10 layers, 36 Rc + 36 Rg periods (four sequences per chain)
    thk = np.array([3., 3, 4, 5, 5, 6, 7, 8, 10, 0]); vs = np.linspace(2.9, 4.6, 10)
    tt = np.arange(5., 41.)
    x0 = np.hstack((vs, thk))
    m = SurfWD(tRc=tt, tRg=tt, device=0)
    d, flag = m.forward(x0); assert flag
    m.set_obsdata(d)
    b0 = bench.bounds_of(x0)
    rng = np.random.default_rng(2)
    xs = np.clip(x0[None, :] * (1 + 0.02 * rng.standard_normal((16, 20))), b0[:, 0], b0[:, 1])
    r = runs_of(m, m._ensure(10), b0, xs, 16, 0.03)
    assert r[0][4][2] > 0
    for i in range(4):
        assert np.array_equal(r[0][i], r[2][i]), i
    _same_counters(r[0][4], r[2][4])


def _same_counters(a, b):
    """Hand-backs and items identical; the stage's evaluation count may differ by the few groups that find their chain already
    handed back by a sister group when they start (which group of a chain runs first is a matter of scheduling)."""
    assert a[0] == b[0] and a[1] == b[1] and a[3] == b[3], (a, b)
    assert abs(a[2] - b[2]) <= 1e-3 * a[2], (a, b)


def test_two_flow_states_in_turn_on_one_context():
    """The warm start belongs to the state whose x array the previous flow call advanced: two states stepped in turn on one
    context start over from the full search at every call (nothing of the other state is continued) and get exactly the
    numbers of the history-free evaluation."""
    import torch
    import bench
    n, nc = 30, 256
    dev = torch.device("cuda")
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    jw, _ = _bench_joint(1); je, _ = _bench_joint(0)
    bounds = bench.bounds_of(bench.true_model(n))
    sts = []
    for seed in (5, 6):
        xs = np.clip(bench.make_models(nc, seed, n), bounds[:, 0], bounds[:, 1])
        st = jw.flow_state(tt(xs), torch.full((nc,), 0.002, dtype=torch.float64, device=dev), tt(bounds))
        st["p"].copy_(tt(0.5 * np.random.default_rng(seed).standard_normal(xs.shape))); st["rem"].fill_(100); st["fresh"].fill_(1)
        sts.append(st)
    ctx = jw._ensure(n)
    i0 = ctx.stat("swd_warm_items")
    for st in sts:
        st["rem"].fill_(3)
    for step in range(4):                        # start evaluation + 3 leapfrog steps, the two states in turn
        for st in sts:
            jw.flow_step(st)
    torch.cuda.synchronize()
    assert ctx.stat("swd_warm_items") == i0      # nothing was continued across the two states
    for st in sts:
        assert int((st["done"] == 1).sum()) == nc
        m, g, d, f = je.misfit_and_grad_device(st["x"].clone())
        okc = (f != 0) & (st["ok"] != 0)
        assert int(okc.sum()) > nc // 2
        assert torch.equal(st["Unew"][okc], m[okc]) and torch.equal(st["dsyn_new"][okc], d[okc])
    # ... while one state alone is
    sts[0]["rem"].fill_(5); sts[0]["fresh"].fill_(1); sts[0]["ok"].fill_(1)
    for _ in range(4):
        jw.flow_step(sts[0])
    torch.cuda.synchronize()
    assert ctx.stat("swd_warm_items") > i0


def test_host_path_restart_in_one_launch_equals_the_scatters():
    """rfs_flow_restart (what run_flow does to chains that go through the host between two steps, in one launch) against
    the dozen torch scatters it replaces: same state afterwards, with and without new step sizes / withdrawn deposits;
    refusals for lists longer than the batch."""
    import torch
    import bench
    from rfsurfhmc_amd._lib import RfsError
    n, nchain = 30, 512
    joint, _ = _bench_joint(1)
    dev = torch.device("cuda")
    rng = np.random.default_rng(3)
    bounds = bench.bounds_of(bench.true_model(n))
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    st = joint.flow_state(tt(bench.make_models(nchain, 1, n)), torch.full((nchain,), 0.01, dtype=torch.float64, device=dev), tt(bounds))
    joint.flow_restart_state(st)
    st["nxt_have"].fill_(1)
    for with_dt, n3 in ((False, 0), (True, 7)):
        idx1 = np.sort(rng.choice(nchain, 40, replace=False)).astype(np.int32)
        idx2 = idx1[rng.random(40) < 0.7]
        idx3 = idx1[:n3]
        xk = rng.standard_normal((len(idx1), 2 * n)); pn = rng.standard_normal((len(idx2), 2 * n))
        rem = rng.integers(5, 21, len(idx2)).astype(np.int32); dtn = rng.random(len(idx2))
        ref = {k: st[k].clone() for k in ("x", "p", "rem", "dt", "fresh", "ok", "nxt_have")}
        ref["x"].index_copy_(0, tt(idx1.astype(np.int64)), tt(xk))
        i2 = tt(idx2.astype(np.int64))
        ref["p"].index_copy_(0, i2, tt(pn)); ref["rem"].index_copy_(0, i2, tt(rem))
        if with_dt:
            ref["dt"].index_copy_(0, i2, tt(dtn))
        ref["fresh"].index_fill_(0, i2, 1); ref["ok"].index_fill_(0, i2, 1)
        if n3:
            ref["nxt_have"].index_fill_(0, tt(idx3.astype(np.int64)), 0)
        # one byte buffer, 8-byte aligned pieces
        parts, off, offs = [], 0, []
        for a in (idx1, xk, idx2, pn, rem, dtn, idx3):
            a = np.ascontiguousarray(a); offs.append(off); parts.append(a); off = (off + a.nbytes + 7) & ~7
        h = np.zeros(off, dtype=np.uint8)
        for o, a in zip(offs, parts):
            h[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)
        buf = tt(h)
        st["ok"].zero_()
        ref["ok"].zero_(); ref["ok"].index_fill_(0, i2, 1)
        joint.flow_restart(st, buf, len(idx1), offs[0], offs[1], len(idx2), offs[2], offs[3], offs[4], offs[5] if with_dt else None,
                           n3, offs[6] if n3 else None)
        torch.cuda.synchronize()
        for k in ref:
            assert torch.equal(st[k], ref[k]), (with_dt, k)
    # lengths and step sizes only (chains the device has restarted already: no momentum, fresh / ok untouched)
    before = {k: st[k].clone() for k in ("x", "p", "fresh", "ok", "nxt_have")}
    joint.flow_restart(st, buf, 0, None, None, len(idx2), offs[2], None, offs[4], offs[5], 0, None)
    torch.cuda.synchronize()
    for k in before:
        assert torch.equal(st[k], before[k]), k
    assert torch.equal(st["rem"][i2], tt(rem)) and torch.equal(st["dt"][i2], tt(dtn))
    with pytest.raises(RfsError):
        joint.flow_restart(st, buf, nchain + 1, 0, 0, 0, None, None, None, None, 0, None)


def test_search_without_a_prediction_for_small_batches(orc):
    """Round 6 ("swd_cold_scan" / "swd_cold_first"): in a batch of a few chains whose hand-backs are searched in the foreground, a
    chain the warm start declines takes k_swd_cold_scan / k_swd_cold_pick -- every period's secular function on one grid, its
    roots refined, the reference's scan replayed on them -- and then the branch test on the reference's own grid and the
    reference-root stage like everybody else, instead of the sequential search.
    (1) Wild models evaluated one after the other (each a different random model: nothing to continue) give the oracle's roots
    and flags; (2) configs[0]'s sampler from its own random start models at dt = 0.1 (one chain of that run hands 40 % of its
    evaluations back) stores the same samples, misfits and accept counts with the stage on, off, and with the warm search left
    out altogether -- and the sequential search runs for less than half as many evaluations."""
    import torch
    import bench
    from rfsurfhmc_amd.model.model_surf import SurfWD
    from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
    thk = np.array([3., 3, 4, 5, 5, 6, 7, 8, 10, 0]); vs = np.linspace(2.9, 4.6, 10)
    tt = np.arange(5., 41.)
    x0 = np.hstack((vs, thk))
    m = SurfWD(tRc=tt, tRg=tt, device=0)
    d0, flag = m.forward(x0); assert flag
    m.set_obsdata(d0)
    b0 = bench.bounds_of(x0)
    ctx = m._ensure(10)
    o = orc.SurfWD(tRc=tt, tRg=tt)
    o.set_obsdata(d0)
    dev = torch.device("cuda")
    names = ("swd_warm_declined_chains", "swd_cold_chains")
    try:
        # (1) nothing to continue: 8 models at a time, every evaluation's models drawn afresh inside the bounds
        m.set_warm_start(2)
        rng = np.random.default_rng(11)
        c0 = [ctx.stat(k) for k in names]
        nroot = nsame = 0
        worst = 0.0
        for it in range(6):
            xs = b0[:, 0] + (b0[:, 1] - b0[:, 0]) * rng.random((8, 20))
            xs[:, 19] = 0.0
            mis, g, d, f = m.misfit_and_grad_device(torch.from_numpy(xs).to(dev))
            d, f, mis, g = d.cpu().numpy(), f.cpu().numpy() != 0, mis.cpu().numpy(), g.cpu().numpy()
            for i in range(8):
                mo, go, do, fo = o.misfit_and_grad(xs[i])
                assert bool(fo) == bool(f[i]), (it, i)
                if not fo:
                    continue
                # (phase velocities are the roots; a root one float32 step off moves a group velocity by ~1e-4 relative)
                rel_c = np.abs(d[i, :36] - do[:36]) / do[:36]
                assert rel_c.max() <= 1.2e-6, (it, i, float(rel_c.max()))
                nroot += 36; nsame += int((d[i, :36] == do[:36]).sum())
                worst = max(worst, float(rel_c.max()))
                if np.array_equal(d[i, :36], do[:36]):
                    assert abs(mis[i] - mo) <= 1e-5 * abs(mo) + 1e-12, (it, i)
        took = [ctx.stat(k) - v for k, v in zip(names, c0)]
        print(f"wild models: {nsame} of {nroot} phase velocities bit-identical, worst {worst:.2e} c; {took[1]} chain evaluations "
              f"through the search without a prediction, {took[0]} through the sequential search")
        # (on models this wild the device's own sequential search ends one float32 step beside the oracle's root for 1.2 % of the
        # roots -- scripts/cold_wild.py: 37 of 3 168 with the stage off, 30 with it on)
        assert took[1] >= 20 and nsame >= 0.97 * nroot
        # (2) the sampler
        m.set_warm_start(1)
        out = {}
        for tag, cold, first, nch in (("off", 0, 0, 1), ("on", -1, 0, 1), ("first", -1, 8, 1), ("off8", 0, 0, 8), ("on8", -1, 0, 8), ("first8", -1, 8, 8)):
            ctx.set_option("swd_cold_scan", cold); ctx.set_option("swd_cold_first", first)
            c0 = [ctx.stat(k) for k in names]
            # (no burn-in: every accepted end point and its misfit is stored, in the order the chain accepted them)
            s = HamitonianMC(m, b0, 0.1, [5, 20], 10, 991206, 800, 0, myrank=0, name="c0", outdir=None, nchains=nch, verbose=False, store_syn=False)
            mis = s.sample_flow(max_steps=240)
            out[tag] = (np.atleast_2d(np.asarray(mis)), np.asarray(s.x_cache), np.asarray(s.naccepted), [ctx.stat(k) - v for k, v in zip(names, c0)])
        for a, b in (("off", "on"), ("off", "first"), ("off8", "on8"), ("off8", "first8")):
            # How many trajectories fit into 240 device steps depends on the host's timing (a restart that comes late costs its chain a
            # step); their results do not.  The accepted end points the two runs have in common, one by one: identical up to the
            # first evaluation in which a root ends a float32 step beside the other run's (on chains this wild about one root in
            # a hundred does, with either search: part 1) -- there the misfits still agree to the contract's 1e-5, afterwards the
            # two trajectories are different trajectories.
            na, nb = out[a][2], out[b][2]
            assert na.sum() > 0 and nb.sum() > 0, (a, b, na, nb)
            nsame = 0
            for c in range(len(na)):
                k = int(min(na[c], nb[c]))
                ma, mb, xa, xb = out[a][0][c, :k], out[b][0][c, :k], out[a][1][c, :k], out[b][1][c, :k]
                diff = (ma != mb) | (xa != xb).any(1)
                if not diff.any():
                    nsame += 1
                    continue
                j = int(np.argmax(diff))
                assert abs(ma[j] - mb[j]) <= 1e-5 * abs(ma[j]), (a, b, c, j, ma[j], mb[j])
                assert np.abs(xa[j] - xb[j]).max() <= 1e-4, (a, b, c, j)
            print(a, b, "chains identical throughout:", nsame, "of", len(na))
        print({k: v[3] for k, v in out.items()})
        assert out["off"][3][0] >= 40                                   # the chain is a wild one
        assert out["on"][3][0] <= 0.7 * out["off"][3][0] and out["first"][3][0] <= 0.5 * out["off"][3][0]
        assert out["on"][3][1] > 0 and out["first8"][3][1] > 0
    finally:
        ctx.set_option("swd_cold_scan", -1); ctx.set_option("swd_cold_first", 8); m.set_warm_start(1)
