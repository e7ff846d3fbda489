"""-m gpu: the batched samplers (device-resident trajectories through rfs_leapfrog_dev) against traces
of the reference's own samplers (tests/golden/sampler_hybrid.npz, produced by oracle/make_golden.py from
the unmodified pyhmc/hmc.py and pyhmc/hmcda.py), and the trajectory kernel against a numpy restatement
of pyhmc/hmc.py:121-190 driven by the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


def _joint(g, hip=True, orc=None, warm=0):
    """warm = 0: every evaluation by the reference-semantics root search -- the mode in which a chain's numbers do not
    depend on its history, so schedules can be compared bit for bit and the reference's traces are met to 1e-6;
    warm = 1: the library's default inside trajectories (tests/test_gpu_warm.py)."""
    t = g["t"]
    if hip:
        from rfsurfhmc_amd.model.model_rf import ReceiverFunc
        from rfsurfhmc_amd.model.model_surf import SurfWD
        from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
        j = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(0.045, 125, 0.4, 1.5, 5.0, 0.001, "P", "freq"), SurfWD(tRc=t, tRg=t))
        j.set_warm_start(warm)
    else:
        j = orc.Joint_RF_SWD(1.0, 1.0, orc.ReceiverFunc(0.045, 125, 0.4, 1.5, 5.0, 0.001, "P", "freq"), orc.SurfWD(tRc=t, tRg=t))
    j.set_obsdata(g["dobs"][:125], g["dobs"][125:])
    return j


def test_hmc_reproduces_reference_ranks_0_and_1(golden):
    """Chain c of a 2-chain sampler == reference MPI rank c: same initial model, L draws, accept
    decisions; states and misfits to 1e-6."""
    from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
    g = golden["sampler_hybrid"]
    joint = _joint(g)
    s = HamitonianMC(joint, g["bounds"], 0.1, [5, 20], 2, 991206, 6, 3, myrank=0, name="t", outdir=None,
                     nchains=2, verbose=False)
    s.trace = []
    mis = s.sample()
    for c, tag in ((0, "hmc_r0"), (1, "hmc_r1")):
        assert np.array_equal(s.initmodel[c], g[f"{tag}/initmodel"])
        seq = [(tr, tr["active"].index(c)) for tr in s.trace if c in tr["active"]]
        L = np.array([tr["L"][k] for tr, k in seq])
        acc = np.array([tr["accept"][k] for tr, k in seq])
        x = np.array([tr["xres"][k] for tr, k in seq])
        U = np.array([tr["Ures"][k] for tr, k in seq])
        assert np.array_equal(L, g[f"{tag}/L"]), (tag, L, g[f"{tag}/L"])
        assert np.array_equal(acc, g[f"{tag}/accept"])
        # U = 1/2 |d - dobs|^2 with residuals of order 1e-2: a 1e-8 deviation of the synthetics (the float32
        # rounding of the Rayleigh roots) shows up as ~1e-6 in U, hence 1e-5 (the north-star bound) for U
        assert rel(x, g[f"{tag}/x"]) < 1e-6 and rel(U, g[f"{tag}/U"]) < 1e-5
        assert rel(mis[c], g[f"{tag}/misfit"]) < 1e-5


def _same_store(mine, ref_path, tol):
    """Member for member: the product's {name}.{rank}.h5 against the file the reference sampler wrote through h5py
    (tests/golden/reference_store, see test_h5_store.py)."""
    from rfsurfhmc_amd.pyhmc import _h5
    with _h5.open_file(mine, "r") as fm, _h5.open_file(ref_path, "r") as fr:
        a = {n: np.asarray(d[...]) for n, d in _h5.walk(fm)}
        b = {n: np.asarray(d[...]) for n, d in _h5.walk(fr)}
    assert set(a) == set(b) and len(a) == 4 + 2 * 6
    for n in b:
        assert a[n].shape == b[n].shape and a[n].dtype == b[n].dtype == np.float64, n
        assert rel(a[n], b[n]) < tol, (n, rel(a[n], b[n]))


@pytest.mark.parametrize("kind", ["hmc", "da"])
def test_sampler_run_lands_in_the_reference_h5_file(kind, golden, tmp_path):
    """End to end on the device path: the same seeded run as the reference's rank 0, written in the reference's HDF5
    layout, equals the file the reference itself wrote (pyhmc/hmc.py:203-226, 272-275): every sample, every synthetic,
    the mean-of-nbest model and its synthetics."""
    import os
    from rfsurfhmc_amd.pyhmc import _h5
    from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
    from rfsurfhmc_amd.pyhmc.hmcda import HMCDualAveraging
    if _h5.backend() is None:
        pytest.skip("neither h5py nor libhdf5 on this box")
    g = golden["sampler_hybrid"]
    joint = _joint(g)
    if kind == "hmc":
        s = HamitonianMC(joint, g["bounds"], 0.1, [5, 20], 2, 991206, 6, 3, myrank=0, name="hmc", outdir=str(tmp_path),
                         nchains=1, verbose=False, store_format="h5")
    else:
        s = HMCDualAveraging(joint, g["bounds"], 0.1, 10, 2, 0.65, 991206, 6, 3, myrank=0, name="da",
                             outdir=str(tmp_path), nchains=1, verbose=False, store_format="h5")
    s.sample()
    assert sorted(os.listdir(tmp_path)) == [f"{kind}.0.h5", f"{kind}.rank0.h5"]
    ref = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_store", f"{kind}.0.h5")
    _same_store(str(tmp_path / f"{kind}.0.h5"), ref, 1e-6)


def test_hmcda_reproduces_reference_rank_0(golden):
    from rfsurfhmc_amd.pyhmc.hmcda import HMCDualAveraging
    g = golden["sampler_hybrid"]
    joint = _joint(g)
    s = HMCDualAveraging(joint, g["bounds"], 0.1, 10, 2, 0.65, 991206, 6, 3, myrank=0, name="t", outdir=None,
                         nchains=1, verbose=False)
    s.trace = []
    mis = s.sample()
    assert np.array_equal(s.initmodel[0], g["da_r0/initmodel"])
    L = np.array([tr["L"][0] for tr in s.trace]); dt = np.array([tr["dt"][0] for tr in s.trace])
    assert np.array_equal(L, g["da_r0/L"])
    assert rel(dt, g["da_r0/dt"]) < 1e-6                       # dual-averaging step sizes
    assert rel(np.array([tr["alpha"][0] for tr in s.trace]), g["da_r0/alpha"]) < 1e-5
    assert rel(np.array([tr["xend"][0] for tr in s.trace]), g["da_r0/x"]) < 1e-6
    assert rel(mis, g["da_r0/misfit"]) < 1e-5


def _ref_leapfrog(model, bounds, x, p0, dt, L):
    """numpy restatement of pyhmc/hmc.py:121-190 (deterministic part)."""
    def mirror(x, p):
        x, p = x.copy(), p.copy()
        hi, lo = bounds[:, 1], bounds[:, 0]
        i1, i2 = x > hi, x < lo
        while np.sum(np.logical_or(i1, i2)) > 0:
            x[i1] = 2 * hi[i1] - x[i1]; p[i1] = -p[i1]
            x[i2] = 2 * lo[i2] - x[i2]; p[i2] = -p[i2]
            i1, i2 = x > hi, x < lo
        return x, p
    p = p0 * 1.0; xn = x * 1.0
    U, grad, dsyn, flag = model.misfit_and_grad(xn)
    if not flag:
        return None
    Hcur = 0.5 * p @ p + U
    p = p - dt * grad * 0.5
    for i in range(L):
        xn = xn + dt * p
        xn, p = mirror(xn, p)
        Un, grad, dn, flag = model.misfit_and_grad(xn)
        if not flag or np.isnan(grad).any():
            return None
        p = p - dt * grad * (1.0 if i < L - 1 else 0.5)
    return xn, Un, Hcur, 0.5 * p @ p + Un, dn


def test_leapfrog_kernel_per_chain_dt_and_L(orc, golden):
    import torch
    g = golden["sampler_hybrid"]
    joint, ojoint = _joint(g), _joint(g, hip=False, orc=orc)
    rng = np.random.default_rng(8)
    bounds = g["bounds"]
    nc, nx = 6, len(g["x0"])
    x = bounds[:, 0] + (bounds[:, 1] - bounds[:, 0]) * rng.random((nc, nx))
    x[:, : nx // 2] = np.sort(x[:, : nx // 2], axis=1)
    x[0] = bounds[:, 1] - 1e-3          # start next to the upper bounds: forces mirror reflections
    p0 = rng.standard_normal((nc, nx)) * 0.5
    dt = np.array([0.1, 0.05, 0.2, 0.1, 0.02, 0.15]); L = np.array([3, 7, 1, 5, 9, 2], dtype=np.int32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    out = joint.leapfrog_device(t(x), t(p0), t(dt), t(L), t(bounds), sort_by_length=True)
    # the length-sorted schedule (each step evaluates only the chains still inside their trajectory) changes nothing
    plain = joint.leapfrog_device(t(x), t(p0), t(dt), t(L), t(bounds), sort_by_length=False)
    for k in out:
        assert torch.equal(out[k], plain[k]), k
    ok = out["ok"].cpu().numpy()
    for c in range(nc):
        ref = _ref_leapfrog(ojoint, bounds, x[c], p0[c], dt[c], int(L[c]))
        assert bool(ok[c]) == (ref is not None)
        if ref is None:
            continue
        xn, Un, Hc, Hn, dn = ref
        assert rel(out["xnew"][c].cpu().numpy(), xn) < 1e-6
        assert abs(out["Unew"][c].item() - Un) < 1e-5 * abs(Un) and abs(out["Hcur"][c].item() - Hc) < 1e-5 * abs(Hc)
        assert abs(out["Hnew"][c].item() - Hn) < 1e-5 * abs(Hn)
        # unsorted "wild" start models: a Rayleigh root may differ by the reference's own 1e-6 c refinement
        # tolerance, which the group velocities amplify (see test_gpu_parity.py)
        assert rel(out["dsyn_new"][c].cpu().numpy(), dn) < 2e-5


def test_flow_schedule_equals_batch_schedule(golden):
    """sample_flow() -- every chain at its own point of its own trajectory, restarted the moment it finishes -- returns
    exactly the samples of sample() (chains are independent and each consumes its own RNG stream in the same order),
    in fewer device steps; chain 0 still reproduces the reference's rank 0."""
    from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
    g = golden["sampler_hybrid"]
    mk = lambda: HamitonianMC(_joint(g), g["bounds"], 0.1, [5, 20], 2, 991206, 6, 3, myrank=0, name="t", outdir=None,
                              nchains=7, verbose=False)
    a = mk(); ma = a.sample()
    a.trace = None
    b = mk(); mb = b.sample_flow()
    assert np.array_equal(ma, mb)
    assert np.array_equal(a.x_cache, b.x_cache) and np.array_equal(a.syndata, b.syndata)
    assert np.array_equal(a.accept_ratio, b.accept_ratio) and np.array_equal(a.xmean, b.xmean)
    assert rel(mb[0], g["hmc_r0/misfit"]) < 1e-5
    # batch: every round costs max(L) + 1 evaluations of every chain; flow: one evaluation per chain and step
    assert b.flow_steps > 0
    # the library's own count of the evaluations the trajectories used (statistic "flow_chain_steps", what bench.py
    # divides by the time): the chains inside a trajectory at every step, counted here on the host side of the same run
    import torch
    c = mk(); seen = []
    ctx = c.model._ensure(len(g["bounds"]) // 2)
    f0 = ctx.stat("flow_chain_steps")
    c.sample_flow(step_hook=lambda s_, st: seen.append(int(((st["rem"] > 0) | (st["fresh"] != 0)).sum().item())))
    n_lib = ctx.stat("flow_chain_steps") - f0
    assert 0 < n_lib <= sum(seen) and n_lib >= sum(seen) - 7 * 3          # (a failed chain idles with rem > 0 until the host takes it)


def test_flow_schedule_equals_batch_schedule_dual_averaging(golden):
    """HMCDualAveraging.sample_flow(): per-chain step sizes give per-chain trajectory lengths; same samples, same
    adapted step sizes as the batch schedule, chain 0 still the reference's rank 0."""
    from rfsurfhmc_amd.pyhmc.hmcda import HMCDualAveraging
    g = golden["sampler_hybrid"]
    mk = lambda: HMCDualAveraging(_joint(g), g["bounds"], 0.1, 10, 2, 0.65, 991206, 6, 3, myrank=0, name="t",
                                  outdir=None, nchains=5, verbose=False)
    a = mk(); ma = a.sample()
    b = mk(); mb = b.sample_flow()
    assert np.array_equal(ma, mb) and np.array_equal(a.x_cache, b.x_cache)
    assert np.array_equal(a.dt_final, b.dt_final) and np.array_equal(a.accept_ratio, b.accept_ratio)
    assert rel(mb[0], g["da_r0/misfit"]) < 1e-5


def test_inverse_mass_matrix(orc, golden):
    """Diagonal inverse mass (rfs_set_inverse_mass): ones reproduce the identity results bit for bit; a non-trivial
    mass follows x' = M^-1 p, K = p.M^-1 p / 2 (numpy restatement of the leapfrog on the oracle); the sampler draws
    p ~ 0.5 N(0, M) and still returns finite, accepted samples."""
    import torch
    from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
    g = golden["sampler_hybrid"]
    joint, ojoint = _joint(g), _joint(g, hip=False, orc=orc)
    rng = np.random.default_rng(18)
    bounds = g["bounds"]
    nc, nx = 4, len(g["x0"])
    x = np.tile(g["x0"], (nc, 1)) * (1 + 0.01 * rng.standard_normal((nc, nx)))
    x[:, -1] = 1.0
    x = np.clip(x, bounds[:, 0] + 1e-6, bounds[:, 1] - 1e-6)
    p0 = rng.standard_normal((nc, nx)) * 0.5
    dt = np.full(nc, 0.02); L = np.array([3, 5, 2, 4], dtype=np.int32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    base = joint.leapfrog_device(t(x), t(p0), t(dt), t(L), t(bounds))
    joint.set_inverse_mass(np.ones(nx))
    ones = joint.leapfrog_device(t(x), t(p0), t(dt), t(L), t(bounds))
    for k in base:
        assert torch.equal(base[k], ones[k]), k
    minv = 0.5 + rng.random(nx)
    joint.set_inverse_mass(minv)
    out = joint.leapfrog_device(t(x), t(p0), t(dt), t(L), t(bounds))
    for c in range(nc):                       # numpy restatement with mass, no reflection happens at these steps
        p = p0[c].copy(); xn = x[c].copy()
        U, grad, _, flag = ojoint.misfit_and_grad(xn)
        Hcur = 0.5 * np.sum(p * p * minv) + U
        p = p - dt[c] * grad * 0.5
        for i in range(L[c]):
            xn = xn + dt[c] * (p * minv)
            assert np.all(xn < bounds[:, 1]) and np.all(xn > bounds[:, 0])
            Un, grad, _, flag = ojoint.misfit_and_grad(xn)
            p = p - dt[c] * grad * (1.0 if i < L[c] - 1 else 0.5)
        assert rel(out["xnew"][c].cpu().numpy(), xn) < 1e-6
        assert abs(out["Hcur"][c].item() - Hcur) < 1e-5 * abs(Hcur)
        assert abs(out["Hnew"][c].item() - (0.5 * np.sum(p * p * minv) + Un)) < 1e-5 * abs(Hcur)
    joint.set_inverse_mass(None)
    again = joint.leapfrog_device(t(x), t(p0), t(dt), t(L), t(bounds))
    assert torch.equal(again["Hnew"], base["Hnew"])
    s = HamitonianMC(_joint(g), bounds, 0.02, [3, 6], 2, 991206, 3, 1, myrank=0, name="t", outdir=None, nchains=3,
                     verbose=False, inverse_mass=minv)
    mis = s.sample_flow(x_init=x[:3])
    assert np.all(np.isfinite(mis)) and np.all(mis > 0)


def test_ensemble_mass_adaptation_on_the_device(golden):
    """mass_adapt on the real joint model: the estimate made from the chains' spread reaches the device
    (rfs_set_inverse_mass), the run equals one that is handed the same M^-1 from the start of that trajectory on,
    and both samplers finish with finite misfits."""
    from rfsurfhmc_amd.pyhmc._batched import ensemble_inverse_mass
    from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
    from rfsurfhmc_amd.pyhmc.hmcda import HMCDualAveraging
    g = golden["sampler_hybrid"]
    bounds = g["bounds"]
    nc, nx = 32, len(g["x0"])
    rng = np.random.default_rng(23)
    x = np.tile(g["x0"], (nc, 1)) * (1 + 0.02 * rng.standard_normal((nc, nx)))
    x[:, -1] = 1.0 + 0.1 * rng.standard_normal(nc)
    x = np.clip(x, bounds[:, 0] + 1e-6, bounds[:, 1] - 1e-6)
    kw = dict(myrank=0, name="t", outdir=None, nchains=nc, verbose=False)
    a = HamitonianMC(_joint(g), bounds, 0.02, [3, 6], 2, 991206, 4, 2, mass_adapt=[0], **kw)
    ma = a.sample(x_init=x)
    minv = ensemble_inverse_mass(x)
    assert np.array_equal(a.inverse_mass, minv) and abs(np.mean(np.log(minv))) < 1e-12
    b = HamitonianMC(_joint(g), bounds, 0.02, [3, 6], 2, 991206, 4, 2, inverse_mass=minv, **kw)
    mb = b.sample(x_init=x)
    assert np.array_equal(ma, mb) and np.array_equal(a.x_cache, b.x_cache)
    assert np.all(np.isfinite(ma)) and a.accept_ratio.mean() > 0.3
    d = HMCDualAveraging(_joint(g), bounds, 0.02, 4, 2, 0.65, 991206, 4, 3, mass_adapt=[0, 2], **kw)
    md = d.sample(x_init=x)
    assert np.all(np.isfinite(md)) and d.inverse_mass is not None and np.all(d.dt_final > 0)
    # the flow schedule adapts at the same trajectory counts (its segments end there): same estimate, same samples as the
    # batch schedule (full search at every step: the two schedules then evaluate identically)
    af = HamitonianMC(_joint(g, warm=0), bounds, 0.02, [3, 6], 2, 991206, 4, 2, mass_adapt=[0, 1], **kw)
    maf = af.sample_flow(x_init=x)
    ab = HamitonianMC(_joint(g, warm=0), bounds, 0.02, [3, 6], 2, 991206, 4, 2, mass_adapt=[0, 1], **kw)
    mab = ab.sample(x_init=x)
    assert np.array_equal(maf, mab) and np.array_equal(af.x_cache, ab.x_cache) and np.array_equal(af.inverse_mass, ab.inverse_mass)
    df = HMCDualAveraging(_joint(g, warm=0), bounds, 0.02, 4, 2, 0.65, 991206, 4, 3, mass_adapt=[0, 2], **kw)
    mdf = df.sample_flow(x_init=x)
    db = HMCDualAveraging(_joint(g, warm=0), bounds, 0.02, 4, 2, 0.65, 991206, 4, 3, mass_adapt=[0, 2], **kw)
    mdb = db.sample(x_init=x)
    assert np.array_equal(mdf, mdb) and np.array_equal(df.dt_final, db.dt_final) and np.array_equal(df.inverse_mass, db.inverse_mass)


def test_flow_step2_accepts_rejects_and_restarts_like_the_host(golden):
    """rfs_flow_step2 at the boundary: two copies of 64 chains take the same trajectories; in one the host does what
    pyhmc/hmc.py:192-198 does after the step that completes them, in the other the draws were deposited beforehand and
    the device did it.  Same models, same momenta, same books, and the deposit is consumed."""
    import torch
    g = golden["sampler_hybrid"]
    joint = _joint(g)
    nc, nx = 64, len(g["x0"])
    rng = np.random.default_rng(3)
    x0 = np.clip(g["x0"][None, :] * (1 + 0.02 * rng.standard_normal((nc, nx))), g["bounds"][:, 0], g["bounds"][:, 1])
    dev = torch.device("cuda")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    L = 3
    p0 = 0.5 * rng.standard_normal((nc, nx)); pnext = 0.5 * rng.standard_normal((nc, nx))
    u = rng.random(nc); u[:8] = 0.0; u[8:16] = 1e300                   # some certain accepts, some certain rejects
    Lnext = rng.integers(2, 6, nc).astype(np.int32)

    def start():
        st = joint.flow_state(t(x0), torch.full((nc,), 0.05, dtype=torch.float64, device=dev), t(g["bounds"]))
        st["p"].copy_(t(p0)); st["rem"].fill_(L); st["fresh"].fill_(1)
        return st
    a, b = start(), joint.flow_restart_state(start())
    for s in range(L + 1):
        if s == L:                                                     # deposits one step before completion
            b["nxt_u"].copy_(t(u)); b["nxt_p"].copy_(t(pnext)); b["nxt_rem"].copy_(t(Lnext)); b["nxt_have"].fill_(1)
            b["nxt_have"][5] = 0                                       # ... except for one chain: it must wait for the host
        joint.flow_step(a); joint.flow_step(b)
    torch.cuda.synchronize()
    assert int(a["done"].sum()) == nc
    Hc, Hn = a["Hcur"].cpu().numpy(), a["Hnew"].cpu().numpy()
    acc = u < np.exp(-(Hn - Hc))
    assert acc[:8].all() and not acc[8:16].any()
    db = b["done"].cpu().numpy()
    want = np.where(acc, 3, 2); want[5] = 1
    assert np.array_equal(db, want)
    dep = np.arange(nc) != 5
    rv = b["res_val"].cpu().numpy()
    assert np.array_equal(rv[dep, 1], Hc[dep]) and np.array_equal(rv[dep, 2], Hn[dep])
    assert np.array_equal(rv[dep, 3], a["Unew"].cpu().numpy()[dep]) and np.array_equal(rv[dep, 0], a["Ucur"].cpu().numpy()[dep])
    xa = a["x"].cpu().numpy()
    assert np.array_equal(b["res_x"].cpu().numpy()[dep], xa[dep])
    xb = b["x"].cpu().numpy()
    assert np.array_equal(xb[dep & acc], xa[dep & acc]) and np.array_equal(xb[dep & ~acc], x0[dep & ~acc])
    assert np.array_equal(b["p"].cpu().numpy()[dep], pnext[dep]) and np.array_equal(b["rem"].cpu().numpy()[dep], Lnext[dep])
    assert np.array_equal(b["fresh"].cpu().numpy(), dep.astype(np.int32)) and int(b["nxt_have"].sum()) == 0
    assert int(b["rem"][5]) == -1 and np.array_equal(xb[5], xa[5])      # the chain without a deposit: rfs_flow_step behaviour
    # next call: the restarted chains evaluate their start model (fresh), the books of the finished trajectory stay parked
    joint.flow_step(b); torch.cuda.synchronize()
    assert np.array_equal(b["res_val"].cpu().numpy(), rv) and int(b["fresh"].sum()) == 0
    assert np.array_equal(b["xstart"].cpu().numpy()[dep], xb[dep]) and np.isinf(b["Hnew"].cpu().numpy()[dep]).all()


@pytest.mark.parametrize("kind", ["hmc", "da"])
def test_flow_with_device_restarts_at_scale_equals_the_synchronous_host_path(kind, golden):
    """2048 chains: restarts on the device, results fetched and deposits made on a side stream beside the running step --
    against the same run with every step followed by a blocking fetch and the host doing the accept / reject
    (pipeline=False, device_restart=False).  Any race between the streams would show as a differing sample."""
    from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
    from rfsurfhmc_amd.pyhmc.hmcda import HMCDualAveraging
    g = golden["sampler_hybrid"]
    nc = 2048 if kind == "hmc" else 256
    rng = np.random.default_rng(21)
    x0 = np.clip(g["x0"][None, :] * (1 + 0.02 * rng.standard_normal((nc, len(g["x0"])))), g["bounds"][:, 0], g["bounds"][:, 1])

    def mk():
        if kind == "hmc":
            return HamitonianMC(_joint(g), g["bounds"], 0.05, [5, 20], 2, 991206, 4, 2, myrank=0, name="t", outdir=None,
                                nchains=nc, verbose=False)
        # (dual averaging lets a chain whose first trajectories are rejected shrink dt, hence lengthen L = lambda / dt: capped)
        return HMCDualAveraging(_joint(g), g["bounds"], 0.05, 10, 2, 0.65, 991206, 3, 12, myrank=0, name="t", outdir=None,
                                nchains=nc, verbose=False, L_cap=30)
    # A start model with a root that equals a layer velocity has the reference's NaN gradient (tests/_refnan.py: here the
    # 1.05 T support root of a group-velocity period of chain 1703 -- the compiled reference returns NaN for it too): every
    # trajectory from it fails at its first step, in the reference as here, and the chain never finishes.  (The root sits ON
    # the layer velocity -- a sign change of the secular function where its formulas switch from oscillatory to evanescent -- and
    # follows it when the model is moved.)  Such chains start from a neighbour's model instead.
    _, g0, _, _ = mk().model.misfit_and_grad(x0)
    stuck = ~np.isfinite(g0).all(axis=1)
    assert stuck.sum() <= 2 and not stuck[0]
    x0[stuck] = x0[0]
    a = mk(); ma = a.sample_flow(x_init=x0, pipeline=False, device_restart=False, max_steps=6000)   # (bounded: never hangs)
    b = mk(); mb = b.sample_flow(x_init=x0, max_steps=6000)
    assert a.finished and b.finished, (a.flow_steps, b.flow_steps)
    assert np.array_equal(ma, mb) and np.array_equal(a.x_cache, b.x_cache) and np.array_equal(a.syndata, b.syndata)
    assert np.array_equal(a.naccepted, b.naccepted) and np.array_equal(a.ntrajectories, b.ntrajectories)
    if kind == "da":
        assert np.array_equal(a.dt_final, b.dt_final)


@pytest.mark.parametrize("kind", ["rf_freq", "rf_time", "swd", "joint_time", "joint_warm"])
def test_flow_schedule_on_every_plugin_kind(kind, golden):
    """The flow entry fuses the drift into the preparation kernel and (frequency-domain RF) the RF reduction into the kick
    kernel; the same samples as the batch schedule must come out for every kind of plugin behind it: RF only (both methods),
    surface waves only, joint with the time-domain RF -- bit for bit with the history-free search -- and, with the warm
    start on, the same accept counts and samples to 1e-4."""
    from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
    from rfsurfhmc_amd.model.model_rf import ReceiverFunc
    from rfsurfhmc_amd.model.model_surf import SurfWD
    from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
    g = golden["sampler_hybrid"]
    t = g["t"]

    def model():
        if kind in ("rf_freq", "rf_time"):
            m = ReceiverFunc(0.045, 125, 0.4, 1.5, 5.0, 0.001, "P", "freq" if kind == "rf_freq" else "time")
            m.set_obsdata(g["dobs"][:125])
        elif kind == "swd":
            m = SurfWD(tRc=t, tRg=t); m.set_warm_start(0)
            m.set_obsdata(g["dobs"][125:])
        else:
            m = Joint_RF_SWD(1.0, 1.0, ReceiverFunc(0.045, 125, 0.4, 1.5, 5.0, 0.001, "P", "time" if kind == "joint_time" else "freq"),
                             SurfWD(tRc=t, tRg=t))
            m.set_warm_start(1 if kind == "joint_warm" else 0)
            m.set_obsdata(g["dobs"][:125], g["dobs"][125:])
        return m
    mk = lambda: HamitonianMC(model(), g["bounds"], 0.05, [3, 9], 2, 991206, 5, 2, myrank=0, name="t", outdir=None,
                              nchains=6, verbose=False)
    a = mk(); ma = a.sample()
    b = mk(); mb = b.sample_flow()
    if kind == "joint_warm":
        assert np.array_equal(a.accept_ratio, b.accept_ratio)
        assert np.all(np.abs(ma - mb) <= 1e-4 * np.abs(ma)) and rel(b.x_cache, a.x_cache) < 1e-4
    else:
        assert np.array_equal(ma, mb) and np.array_equal(a.x_cache, b.x_cache) and np.array_equal(a.accept_ratio, b.accept_ratio)
    assert np.all(np.isfinite(mb)) and b.flow_steps > 0
