"""Ensemble mass adaptation: M^-1 from the cross-chain variance, applied between trajectories of the
batch schedule, carried through checkpoints, pooled over ranks.  TOY model on CPU tensors (host logic only)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rfsurfhmc_amd.pyhmc._batched import ensemble_inverse_mass


class AnisoToy:
    """U(x) = 0.5 sum(((x - m) / s)^2) with very different scales s; leapfrog with a diagonal inverse mass:
    drift x += dt M^-1 p, K = p.M^-1 p / 2 (what rfs_set_inverse_mass makes the device do)."""
    torch_device = torch.device("cpu")

    def __init__(self, s):
        self.s = np.asarray(s, dtype=float)
        self.m = np.zeros(len(s))
        self.dobs = self.m.copy()
        self.minv = np.ones(len(s))
        self.mass_calls = 0

    def set_inverse_mass(self, minv):
        self.minv = np.ones(len(self.s)) if minv is None else np.asarray(minv, dtype=float).copy()
        self.mass_calls += 1

    def misfit_and_grad(self, x):
        x = np.atleast_2d(x)
        r = (x - self.m) / self.s
        return 0.5 * np.sum(r * r, axis=1), r / self.s, x.copy(), np.ones(len(x), bool)

    def leapfrog_device(self, x, p, dt, L, bounds):
        x = x.clone(); p = p.clone()
        s2 = torch.from_numpy(self.s ** 2); mi = torch.from_numpy(self.minv)
        U = lambda x: 0.5 * (x * x / s2).sum(1)
        K = lambda p: 0.5 * (p * p * mi).sum(1)
        U0 = U(x); H0 = U0 + K(p); x0 = x.clone()
        xn, Un, Hn = x.clone(), U0.clone(), H0.clone()
        p = p - 0.5 * dt[:, None] * x / s2
        for k in range(int(L.max())):
            live = (L > k)[:, None]
            x = torch.where(live, x + dt[:, None] * p * mi, x)
            g = x / s2
            done = L == k + 1
            p = torch.where(live, p - torch.where(done[:, None], 0.5, 1.0) * dt[:, None] * g, p)
            xn = torch.where(done[:, None], x, xn); Un = torch.where(done, U(x), Un)
            Hn = torch.where(done, U(x) + K(p), Hn)
        return dict(ok=torch.ones(len(x), dtype=torch.int32), Hcur=H0, Hnew=Hn, xnew=xn, Unew=Un, Ucur=U0,
                    dsyn_new=xn.clone(), dsyn_cur=x0)


S = np.array([0.02, 0.1, 0.5, 1.0, 2.0, 5.0])


def _bounds(n):
    return np.stack([np.full(n, -100.0), np.full(n, 100.0)], axis=1)


def _make(kind, tmp, nchains=64, **kw):
    from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
    from rfsurfhmc_amd.pyhmc.hmcda import HMCDualAveraging
    common = dict(myrank=0, name="aniso", outdir=str(tmp), nchains=nchains, verbose=False, store_syn=False,
                  per_chain_files=False, **kw)
    if kind == "hmc":
        return HamitonianMC(AnisoToy(S), _bounds(6), 0.5, [4, 10], 4, 7, 30, 30, **common)
    return HMCDualAveraging(AnisoToy(S), _bounds(6), 0.5, 6, 4, 0.65, 7, 30, 30, **common)


def test_ensemble_inverse_mass_recovers_the_scale_ratios():
    x = np.random.default_rng(0).normal(size=(20000, 6)) * S
    minv = ensemble_inverse_mass(x)
    assert abs(np.mean(np.log(minv))) < 1e-12                      # geometric mean 1
    want = S ** 2 / np.exp(np.mean(np.log(S ** 2)))
    assert np.allclose(minv, want, rtol=0.05)
    flat = ensemble_inverse_mass(np.ones((8, 3)))                  # chains that do not differ: identity
    assert np.array_equal(flat, np.ones(3))
    part = ensemble_inverse_mass(np.array([[0.0, 1.0], [0.0, 3.0]]))
    assert part[0] == 1e-3 and part[1] == 1.0                      # zero-variance parameter -> lower clip


def test_adaptation_rescues_an_ill_scaled_posterior(tmp_path):
    """With M = I and dt = 0.5 the narrowest direction (s = 0.02) is far outside leapfrog stability: nothing is
    accepted from the typical set.  After one ensemble estimate the same dt is stable in every direction."""
    x0 = np.random.default_rng(1).normal(size=(64, 6)) * S
    plain = _make("hmc", tmp_path / "p")
    plain.sample(x_init=x0, max_trajectories=12)
    plain_acc = plain.ii / (64 * 12)
    ad = _make("hmc", tmp_path / "a", mass_adapt=[0])
    ad.sample(x_init=x0, max_trajectories=12)
    ad_acc = ad.ii / (64 * 12)
    assert ad.model.mass_calls == 1
    assert np.allclose(ad.inverse_mass, S ** 2 / np.exp(np.mean(np.log(S ** 2))), rtol=0.6)
    assert plain_acc < 0.05 and ad_acc > 0.6, (plain_acc, ad_acc)


@pytest.mark.parametrize("kind", ["hmc", "hmcda"])
def test_resume_carries_the_adapted_mass(kind, tmp_path):
    x0 = np.random.default_rng(2).normal(size=(16, 6)) * S
    full = _make(kind, tmp_path / "f", nchains=16, mass_adapt=[0, 3, 9])
    mis_full = full.sample(x_init=x0)
    assert full.model.mass_calls == 3
    ck = str(tmp_path / "state.npz")
    part = _make(kind, tmp_path / "r", nchains=16, mass_adapt=[0, 3, 9], checkpoint=ck)
    part.sample(x_init=x0, max_trajectories=5)                      # stops between the 2nd and 3rd adaptation
    assert not part.finished
    rest = _make(kind, tmp_path / "r", nchains=16, mass_adapt=[0, 3, 9], checkpoint=ck)
    mis = rest.sample(resume=True)
    assert rest.finished and rest.model.mass_calls == 2             # restored from the file + the one at 9
    assert np.array_equal(mis, mis_full)
    assert np.array_equal(rest.inverse_mass, full.inverse_mass)
    assert np.array_equal(rest.x_cache, full.x_cache)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rfsurfhmc_amd.chains import pooled_variance, shard_range
        x = np.random.default_rng(5).normal(size=(101, 4)) * np.array([1.0, 2.0, 3.0, 0.5]) + 7.0
        a, b = shard_range(101, rank, world)
        q.put((rank, pooled_variance(x[a:b]), ensemble_inverse_mass(x[a:b])))
    finally:
        dist.destroy_process_group()


def test_two_rank_pooled_variance_equals_the_single_process_one():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    x = np.random.default_rng(5).normal(size=(101, 4)) * np.array([1.0, 2.0, 3.0, 0.5]) + 7.0
    for _, var, minv in got:
        assert np.allclose(var, x.var(axis=0), rtol=1e-12)
        assert np.allclose(minv, ensemble_inverse_mass(x), rtol=1e-12)
    assert np.array_equal(got[0][1], got[1][1])                     # every rank ends with the same M
