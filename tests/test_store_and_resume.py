"""Host logic of the batched samplers that needs no GPU: result store (batched file, per-chain exporter with the
reference's member names) and checkpoint / resume.  The model here is a TOY stand-in with the plugin interface
(dobs, misfit_and_grad, leapfrog_device on CPU tensors) -- it exercises the samplers' bookkeeping, not the HIP path."""
import os
import sys

import numpy as np
import pytest
import torch


class ToyModel:
    """U(x) = 0.5 |x - m|^2, synthetics = x; leapfrog with the reference's half-step scheme, no reflection."""
    torch_device = torch.device("cpu")

    def __init__(self, n, fail_above=None):
        self.m = np.linspace(1.0, 2.0, n)
        self.dobs = self.m.copy()
        self.fail_above = fail_above          # an evaluation with U above this "fails" (the reference's flag False)

    def misfit_and_grad(self, x):
        x = np.atleast_2d(x)
        r = x - self.m
        return 0.5 * np.sum(r * r, axis=1), r, x.copy(), np.ones(len(x), bool)

    def leapfrog_device(self, x, p, dt, L, bounds):
        x = x.clone(); p = p.clone()
        m = torch.from_numpy(self.m)
        U0 = 0.5 * ((x - m) ** 2).sum(1); H0 = U0 + 0.5 * (p * p).sum(1)
        x0 = x.clone()
        Lmax = int(L.max())
        xn, Un, Hn = x.clone(), U0.clone(), H0.clone()
        p = p - 0.5 * dt[:, None] * (x - m)
        bad = torch.zeros(len(x), dtype=torch.bool) if self.fail_above is None else U0 > self.fail_above
        for s in range(Lmax):
            live = (L > s)[:, None] & ~bad[:, None]
            x = torch.where(live, x + dt[:, None] * p, x)
            g = x - m
            if self.fail_above is not None:
                bad = bad | (live[:, 0] & (0.5 * (g * g).sum(1) > self.fail_above))
                live = live & ~bad[:, None]
            last = (L == s + 1)[:, None]
            p = torch.where(live, p - torch.where(last, 0.5, 1.0) * dt[:, None] * g, p)
            done = (L == s + 1)
            U = 0.5 * (g * g).sum(1)
            xn = torch.where(done[:, None], x, xn); Un = torch.where(done, U, Un)
            Hn = torch.where(done, U + 0.5 * (p * p).sum(1), Hn)
        return dict(ok=(~bad).to(torch.int32), Hcur=H0, Hnew=Hn, xnew=xn, Unew=Un, Ucur=U0,
                    dsyn_new=xn.clone(), dsyn_cur=x0)


def _toy_flow_state(self, x0, dt, bounds):
    n, nx = x0.shape
    z = lambda *sh: torch.zeros(*sh, dtype=torch.float64)
    return dict(x=x0.clone(), p=z(n, nx), dt=dt.clone(), rem=torch.full((n,), -1, dtype=torch.int32),
                fresh=torch.zeros(n, dtype=torch.int32), bounds=bounds, Ucur=z(n), Hcur=z(n), Unew=z(n), Hnew=z(n),
                dsyn_cur=z(n, nx), dsyn_new=z(n, nx), ok=torch.ones(n, dtype=torch.int32),
                done=torch.zeros(n, dtype=torch.int32))


def _toy_flow_step(self, st):
    """rfs_flow_step semantics (include/rfsurf.h) for the toy potential, same arithmetic as leapfrog_device above."""
    m = torch.from_numpy(self.m)
    x, p, dt, rem, fresh = st["x"], st["p"], st["dt"], st["rem"], st["fresh"]
    run = (fresh == 0) & (rem > 0) & (st["ok"] == 1)
    defer = st.get("kick") is not None
    if defer:                                    # the half kick a deferred start left open
        kk = run & (st["kick"] == 1)
        p[kk] = p[kk] - 0.5 * dt[kk, None] * st["gsave"][kk]
        st["kick"][run] = 0
    x[run] = x[run] + dt[run, None] * p[run]
    g = x - m
    U = 0.5 * (g * g).sum(1)
    st["done"].zero_()
    fr = fresh == 1
    fail = torch.zeros_like(run) if self.fail_above is None else (U > self.fail_above)
    if fr.any():
        st["Ucur"][fr] = U[fr]; st["Unew"][fr] = U[fr]
        st["Hcur"][fr] = U[fr] + 0.5 * (p[fr] * p[fr]).sum(1)
        st["Hnew"][fr] = float("inf")
        st["dsyn_cur"][fr] = x[fr]; st["dsyn_new"][fr] = x[fr]
        if defer:
            st["gsave"][fr] = g[fr]; st["kick"][fr] = (~fail[fr]).to(torch.int32)
        else:
            p[fr] = p[fr] - 0.5 * dt[fr, None] * g[fr]
        fresh[fr] = 0
        st["ok"][fr] = (~fail[fr]).to(torch.int32)
        ff = fr & fail
        rem[ff] = -1; st["done"][ff] = 1
    rf = run & fail                              # a running chain whose evaluation fails: no kick, the host restarts it
    if rf.any():
        st["ok"][rf] = 0; rem[rf] = -1; st["done"][rf] = 1
        run = run & ~fail
    if "nxt_have" in st and fr.any():
        st["xstart"][fr] = x[fr]
    if run.any():
        last = run & (rem == 1)
        w = torch.where(last, 0.5, 1.0)
        p[run] = p[run] - (w[run] * dt[run])[:, None] * g[run]
        st["Unew"][last] = U[last]
        st["Hnew"][last] = U[last] + 0.5 * (p[last] * p[last]).sum(1)
        st["dsyn_new"][last] = x[last]
        rem[run] = rem[run] - 1
        rem[last] = -1
        st["done"][last] = 1
        if "nxt_have" in st:                     # rfs_flow_step2: accept / reject and restart where a deposit waits
            au = last & (st["nxt_have"] == 1)
            if au.any():
                acc = au & (st["nxt_u"] < torch.exp(-(st["Hnew"] - st["Hcur"])))
                st["res_val"][au] = torch.stack([st["Ucur"], st["Hcur"], st["Hnew"], st["Unew"]], dim=1)[au]
                st["res_x"][au] = x[au]
                if st.get("res_dsyn") is not None:
                    st["res_dsyn"][au] = x[au]
                rej = au & ~acc
                x[rej] = st["xstart"][rej]
                p[au] = st["nxt_p"][au]; fresh[au] = 1
                rem[au] = st["nxt_rem"][au] if st.get("nxt_rem") is not None else (1 << 30)
                st["done"][au] = 2; st["done"][acc] = 3
                st["nxt_have"][au] = 0


def _toy_flow_restart_state(self, st, want_dsyn=False, deferred=False):
    n, nx = st["x"].shape
    z = lambda *sh: torch.zeros(*sh, dtype=torch.float64)
    st.update(nxt_have=torch.zeros(n, dtype=torch.int32), nxt_u=z(n), nxt_p=z(n, nx), nxt_rem=torch.zeros(n, dtype=torch.int32),
              xstart=st["x"].clone(), res_x=z(n, nx), res_val=z(n, 4), res_dsyn=z(n, nx) if want_dsyn else None,
              gsave=None, kick=None)
    if deferred:
        st.update(nxt_rem=None, gsave=z(n, nx), kick=torch.zeros(n, dtype=torch.int32))
    return st


ToyModel.flow_state = _toy_flow_state
ToyModel.flow_step = _toy_flow_step
ToyModel.flow_restart_state = _toy_flow_restart_state


def _bounds(n):
    return np.stack([np.full(n, -5.0), np.full(n, 5.0)], axis=1)


def _make(kind, tmp, fail_above=None, **kw):
    from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
    from rfsurfhmc_amd.pyhmc.hmcda import HMCDualAveraging
    n = 6
    common = dict(myrank=1, name="toy", outdir=None if tmp is None else str(tmp), nchains=kw.pop("nchains", 5), verbose=False, **kw)
    if kind == "hmc":
        return HamitonianMC(ToyModel(n, fail_above), _bounds(n), 0.3, [3, 8], 4, 991206, 12, 4, **common)
    return HMCDualAveraging(ToyModel(n, fail_above), _bounds(n), 0.3, 5, 4, 0.65, 991206, 12, 4, **common)


@pytest.mark.parametrize("kind", ["hmc", "hmcda"])
def test_resume_reproduces_an_uninterrupted_run(kind, tmp_path):
    x0 = np.random.default_rng(3).uniform(0, 3, (5, 6))
    full = _make(kind, tmp_path / "a")
    mis_full = full.sample(x_init=x0)
    ck = str(tmp_path / "state.npz")
    part = _make(kind, tmp_path / "b", checkpoint=ck, checkpoint_every=2)
    part.sample(x_init=x0, max_trajectories=7)
    assert not part.finished and os.path.exists(ck)
    rest = _make(kind, tmp_path / "b", checkpoint=ck)          # a fresh process: new object, same arguments
    mis = rest.sample(resume=True)
    assert rest.finished
    assert np.array_equal(mis, mis_full)
    assert np.array_equal(rest.x_cache, full.x_cache) and np.array_equal(rest.xmean, full.xmean)
    from rfsurfhmc_amd.pyhmc._batched import load_batched_results
    a = load_batched_results(full.result_file); b = load_batched_results(rest.result_file)
    assert set(a) == set(b)
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def _formats():
    from rfsurfhmc_amd.pyhmc import _h5
    return ["npz", "h5"] if _h5.backend() else ["npz"]


@pytest.mark.parametrize("fmt", ["npz", "h5"])
def test_result_store_layouts(fmt, tmp_path):
    from rfsurfhmc_amd.pyhmc._batched import export_chain, load_batched_results, load_chain_results
    if fmt not in _formats():
        pytest.skip("neither h5py nor libhdf5 here")
    s = _make("hmc", tmp_path, store_format=fmt)
    mis = s.sample()
    z = load_batched_results(s.result_file)
    assert os.path.basename(s.result_file) == f"toy.rank1.{fmt}" and int(z["first_chain"]) == 5
    assert z["model"].shape == (5, 12, 6) and z["syn"].shape == (5, 12, 6) and np.array_equal(z["misfit"], mis)
    # per-chain files (written by default for a few chains): the reference's members
    for c in range(5, 10):
        f = load_chain_results(str(tmp_path / f"toy.{c}.{fmt}"))
        assert set(f) == {"initmodel", "obs", "mean/model", "mean/syn", "model", "syn"}
        assert np.array_equal(f["model"], z["model"][c - 5]) and np.array_equal(f["mean/model"], z["mean_model"][c - 5])
        assert np.array_equal(f["initmodel"], z["initmodel"][c - 5]) and np.array_equal(f["obs"], z["obs"])
    # exporter recreates one chain's file from the batched one, in either format
    os.remove(tmp_path / f"toy.7.{fmt}")
    p = export_chain(s.result_file, 7, fmt=fmt)
    assert p.endswith(f"toy.7.{fmt}") and np.array_equal(load_chain_results(p)["syn"], z["syn"][2])
    other = "npz" if fmt == "h5" else _formats()[-1]
    q = export_chain(s.result_file, 8, outdir=str(tmp_path / "x"), fmt=other)
    assert np.array_equal(load_chain_results(q)["model"], z["model"][3])
    with pytest.raises(IndexError):
        export_chain(s.result_file, 99)
    # many chains: batched file only
    big = _make("hmc", tmp_path / "big", per_chain_files=False, store_format=fmt)
    big.sample()
    assert sorted(os.listdir(tmp_path / "big")) == [f"toy.rank1.{fmt}"]


def test_store_format_auto_and_errors(tmp_path, monkeypatch):
    from rfsurfhmc_amd.pyhmc import _batched, _h5
    assert _batched.store_format("auto") == _formats()[-1] and _batched.store_format("npz") == "npz"
    with pytest.raises(ValueError):
        _batched.store_format("hdf")
    # no HDF5 at all: auto falls back to npz, an explicit "h5" says what it looked for
    monkeypatch.setattr(_h5, "_lib", None)
    monkeypatch.setenv("RFSURF_HDF5_LIB", str(tmp_path / "nowhere.so"))
    monkeypatch.setitem(sys.modules, "h5py", None)
    assert _h5.backend() is None and _batched.store_format("auto") == "npz"
    with pytest.raises(ImportError, match="RFSURF_HDF5_LIB"):
        _batched.store_format("h5")


@pytest.mark.parametrize("kind", ["hmc", "hmcda"])
def test_flow_schedule_bookkeeping_equals_batch(kind, tmp_path):
    """sample_flow(): chains restarted one by one as they finish (continuous flow) give the samples of the batch
    schedule -- exercised here on the toy model; the GPU suite repeats it on the HIP path."""
    x0 = np.random.default_rng(5).uniform(0, 3, (5, 6))
    a = _make(kind, tmp_path / "a"); ma = a.sample(x_init=x0)
    b = _make(kind, tmp_path / "b"); mb = b.sample_flow(x_init=x0)
    c = _make(kind, tmp_path / "c"); mc = c.sample_flow(x_init=x0, pipeline=False)
    assert np.array_equal(ma, mb) and np.array_equal(ma, mc) and c.flow_steps <= b.flow_steps
    # restarts on the device (early draws) against restarts by the host: same samples, fewer steps
    d = _make(kind, tmp_path / "d"); md = d.sample_flow(x_init=x0, device_restart=False)
    assert np.array_equal(ma, md) and np.array_equal(a.x_cache, d.x_cache) and b.flow_steps < d.flow_steps
    assert np.array_equal(b.accept_ratio, d.accept_ratio)
    assert np.array_equal(a.x_cache, b.x_cache) and np.array_equal(a.syndata, b.syndata)
    assert np.array_equal(a.accept_ratio, b.accept_ratio)
    if kind == "hmcda":
        assert np.array_equal(a.dt_final, b.dt_final)
    assert b.flow_steps > 0


def test_host_threads_caps_and_restores():
    """The samplers' host side runs with torch's intra-op pool capped (a cgroup CPU quota below the visible CPU count
    otherwise throttles the process, see _batched.host_threads) and gives the setting back."""
    import torch
    from rfsurfhmc_amd.pyhmc._batched import cpu_quota, host_threads
    before = torch.get_num_threads()
    assert 1 <= cpu_quota() <= (os.cpu_count() or 1)
    with host_threads() as cap:
        assert cap == max(1, min(before, 4, cpu_quota())) and torch.get_num_threads() == cap
        with host_threads(1) as inner:
            assert inner == 1 and torch.get_num_threads() == 1
        assert torch.get_num_threads() == cap
    assert torch.get_num_threads() == before
    s = _make("hmc", None)
    s.outdir = None
    s.sample()
    assert torch.get_num_threads() == before


@pytest.mark.parametrize("kind", ["hmc", "hmcda"])
def test_device_restarts_with_failing_trajectories_keep_the_reference_draw_order(kind):
    """Trajectories that fail skip the acceptance draw (hmc.py:156,173,177,179).  With restarts on the device that draw
    and the next L and momentum are made one step before a trajectory completes -- so a chain that fails in its last
    two steps has drawn too much, and its stream must be rewound.  40 chains of which a good part fail now and then:
    batch schedule, flow with host restarts and flow with device restarts give the same samples and the same counts."""
    # start models near the minimum (a chain whose START model fails can never leave it, in the reference either); the
    # threshold is crossed by the excursions of the more energetic trajectories only
    x0 = np.linspace(1.0, 2.0, 6)[None, :] + 0.2 * np.random.default_rng(11).standard_normal((40, 6))
    runs = {}
    for name, kw in (("batch", None), ("flow_host", dict(device_restart=False)), ("flow_dev", dict(device_restart=True)),
                     ("flow_dev_sync", dict(device_restart=True, pipeline=False))):
        smp = _make(kind, None, fail_above=1.0, nchains=40)
        mis = smp.sample(x_init=x0) if kw is None else smp.sample_flow(x_init=x0, **kw)
        runs[name] = (mis, smp.x_cache.copy(), smp.accept_ratio.copy(), smp)
    base = runs["batch"]
    assert base[2].min() < 0.8 and base[2].max() <= 1.0           # rejections and failures did happen
    for name in ("flow_host", "flow_dev", "flow_dev_sync"):
        assert np.array_equal(runs[name][0], base[0]), name
        assert np.array_equal(runs[name][1], base[1]) and np.array_equal(runs[name][2], base[2]), name
    assert runs["flow_dev"][3].flow_steps < runs["flow_host"][3].flow_steps
    if kind == "hmc":
        assert runs["flow_dev"][3].flow_withdrawn > 0 and runs["flow_dev_sync"][3].flow_withdrawn > 0  # the rewind was exercised


@pytest.mark.parametrize("kind", ["hmc", "hmcda"])
def test_flow_schedule_checkpoints_at_trajectory_counts(kind, tmp_path):
    """sample_flow with checkpoint_every: segments that end when every chain has completed that many more trajectories.
    The uninterrupted flow run, a run cut inside a later segment and resumed from its last barrier, the batch schedule,
    and a checkpoint written by one schedule and continued by the other: the same samples, bit for bit."""
    x0 = np.random.default_rng(2).normal(size=(5, 6))
    ref = _make(kind, tmp_path / "r"); mr = ref.sample(x_init=x0)
    full = _make(kind, tmp_path / "a", checkpoint=str(tmp_path / "a.npz"), checkpoint_every=3)
    mf = full.sample_flow(x_init=x0)
    assert full.finished and np.array_equal(mf, mr) and np.array_equal(full.x_cache, ref.x_cache)
    ck = str(tmp_path / "b.npz")
    part = _make(kind, tmp_path / "b", checkpoint=ck, checkpoint_every=3)
    part.sample_flow(x_init=x0, max_steps=int(0.6 * full.flow_steps))
    assert not part.finished and os.path.exists(ck)
    rest = _make(kind, tmp_path / "b", checkpoint=ck, checkpoint_every=3)
    mres = rest.sample_flow(resume=True)
    assert rest.finished and np.array_equal(mres, mr) and np.array_equal(rest.x_cache, ref.x_cache)
    ck2 = str(tmp_path / "c.npz")
    p2 = _make(kind, tmp_path / "c", checkpoint=ck2, checkpoint_every=3)
    p2.sample(x_init=x0, max_trajectories=6)
    assert not p2.finished
    r2 = _make(kind, tmp_path / "c", checkpoint=ck2, checkpoint_every=3)
    m2 = r2.sample_flow(resume=True)
    assert r2.finished and np.array_equal(m2, mr)
    r3 = _make(kind, tmp_path / "b2", checkpoint=ck, checkpoint_every=3)
    m3 = r3.sample(resume=True)                   # ... and the flow's checkpoint by the batch schedule
    assert r3.finished and np.array_equal(m3, mr)
