"""HamitonianMC -- batched mirror of the reference sampler pyhmc/hmc.py.

Same constructor arguments, ``init(model, boundaries, rank, **hmc_block)`` keys and ``sample()``
contract; additionally ``nchains`` chains advance together, each trajectory running on the GPU
through rfs_leapfrog_dev (leapfrog + mirror + misfit/gradient, nothing returns to the host
inside a trajectory).  Chain c of a sampler created with ``myrank=r`` reproduces the reference's
MPI rank ``r*nchains + c``: same RNG stream (seed + rank), same draw order per iteration --
randint(L), randn(n)*0.5, [trajectory], rand() -- and the rand() is skipped when the trajectory
fails, as in the reference (hmc.py:156,173,177,179)."""
import sys

import numpy as np

from ._batched import (ChainRNG, ensemble_inverse_mass, initial_models, load_checkpoint, run_flow, save_batched_results,
                       save_chain_results, save_checkpoint)
from ._batched import store_format as resolve_store_format, with_host_threads


class HamitonianMC:
    def __init__(self, UserDefinedModel, boundaries, dt, Lrange, nbest_model, seed, nsamples, ndraws,
                 myrank=0, name="mychain", outdir="./", nchains=1, store_syn=True, verbose=True,
                 per_chain_files=None, checkpoint=None, checkpoint_every=0, inverse_mass=None, mass_adapt=None,
                 store_format="auto", warm_start=None):
        self.myrank = myrank
        self.nchains = int(nchains)
        self.first_chain = myrank * self.nchains
        self.seed = seed
        self.boundaries = np.asarray(boundaries, dtype=np.float64)
        self.Lrange = Lrange
        self.dt = dt
        self.model = UserDefinedModel
        self.nbest_model = nbest_model
        self.nsamples = nsamples
        self.ndraws = ndraws
        self.name, self.outdir = name, outdir
        self.store_syn, self.verbose = store_syn, verbose
        # result files: "h5" = the reference's HDF5 (pyhmc/hmc.py:58,203-226), "npz", or "auto" = HDF5 where h5py or
        # libhdf5 is present; resolved when the files are written so that a missing library cannot fail a finished run early
        self.store_format = store_format
        # result files: one per chain with the reference's names (default for a few chains) and / or one
        # batched file per rank (always written when outdir is set); checkpoint: path of a resumable state file
        self.per_chain_files = (self.nchains <= 16) if per_chain_files is None else per_chain_files
        self.checkpoint, self.checkpoint_every = checkpoint, int(checkpoint_every)
        # warm_start: None = the plugin's setting (library default: on); 0 / False = every evaluation by the
        # reference-semantics root search (rfs_set_option "swd_warm_start")
        if warm_start is not None and hasattr(self.model, "set_warm_start"):
            self.model.set_warm_start(int(warm_start))
        # diagonal inverse mass M^-1 (None = the reference's identity): momenta are drawn as 0.5 z sqrt(M), the
        # device drifts with M^-1 p and uses K = p.M^-1 p / 2 (rfs_set_inverse_mass)
        self.inverse_mass = None if inverse_mass is None else np.asarray(inverse_mass, dtype=np.float64)
        self._pscale = 0.5 if self.inverse_mass is None else 0.5 / np.sqrt(self.inverse_mass)
        # mass_adapt: trajectory counts (inside the burn-in) at which M^-1 is re-estimated from the cross-chain
        # variance of the current models (ensemble_inverse_mass); batch schedule only
        self.mass_adapt = None if mass_adapt is None else frozenset(int(k) for k in mass_adapt)
        # every rank must reach every adaptation point (they are collective): no chain can finish before ndraws
        # trajectories, so points below ndraws are safe on every rank whatever its acceptance rate
        if self.mass_adapt and max(self.mass_adapt) >= ndraws:
            raise ValueError(f"mass_adapt points must lie inside the burn-in (< ndraws = {ndraws}): {sorted(self.mass_adapt)}")
        self.rng = ChainRNG(seed, self.first_chain, self.nchains)
        self.ii = 0
        self.trace = None          # optional list collecting per-iteration records (tests)

    @classmethod
    def init(cls, UserDefinedModel, boundaries, rank, **kargs):
        """pyhmc/hmc.py:63-72 (+ optional keys ``nchains``, ``mass_adapt``)."""
        return cls(UserDefinedModel, boundaries, kargs["dt"], kargs["Lrange"], kargs["nbest"], kargs["seed"],
                   kargs["nsamples"], kargs["ndraws"], rank, kargs["name"], kargs["OUTPUT_DIR"],
                   nchains=kargs.get("nchains", 1), mass_adapt=kargs.get("mass_adapt"),
                   checkpoint=kargs.get("checkpoint"), checkpoint_every=kargs.get("checkpoint_every", 0),
                   store_format=kargs.get("store_format", "auto"), warm_start=kargs.get("warm_start"))

    def _set_inverse_mass(self, minv):
        self.inverse_mass = np.asarray(minv, dtype=np.float64)
        self._pscale = 0.5 / np.sqrt(self.inverse_mass)
        self.model.set_inverse_mass(self.inverse_mass)

    def _device(self):
        import torch
        dev = getattr(self.model, "torch_device", None)       # host-logic tests plug in a CPU model here
        return dev if dev is not None else torch.device("cuda", getattr(self.model, "device", 0) or 0)

    def _leapfrog(self, x, active, L):
        """One trajectory for the chains in ``active`` (pyhmc/hmc.py:140-201).  Returns per-chain
        (xnew, U, dsyn, accept) with the reference's failure return (xcur, inf, dobs, False)."""
        import torch
        dev = self._device()
        n = x.shape[1]
        p0 = self.rng.randn(active, n) * self._pscale                                  # hmc.py:146
        xd = torch.from_numpy(np.ascontiguousarray(x[active])).to(dev)
        pd = torch.from_numpy(np.ascontiguousarray(p0)).to(dev)
        dtd = torch.full((len(active),), float(self.dt), dtype=torch.float64, device=dev)
        Ld = torch.from_numpy(np.ascontiguousarray(L)).to(dev)
        bd = torch.from_numpy(np.ascontiguousarray(self.boundaries)).to(dev)
        out = self.model.leapfrog_device(xd, pd, dtd, Ld, bd)
        ok = out["ok"].cpu().numpy().astype(bool)
        Hcur, Hnew = out["Hcur"].cpu().numpy(), out["Hnew"].cpu().numpy()
        xnew, Unew = out["xnew"].cpu().numpy(), out["Unew"].cpu().numpy()
        Ucur = out["Ucur"].cpu().numpy()
        dnew, dcur = out["dsyn_new"].cpu().numpy(), out["dsyn_cur"].cpu().numpy()
        u = np.full(len(active), np.nan)
        u[ok] = self.rng.rand([active[i] for i in np.nonzero(ok)[0]])        # hmc.py:193, skipped on failure
        with np.errstate(over="ignore", invalid="ignore"):
            accept = ok & (u < np.exp(-(Hnew - Hcur)))
        xres = np.where(accept[:, None], xnew, x[active])
        Ures = np.where(accept, Unew, np.where(ok, Ucur, np.inf))
        dres = np.where(accept[:, None], dnew, np.where(ok[:, None], dcur, self.model.dobs[None, :]))
        if self.trace is not None:
            self.trace.append(dict(active=list(active), L=L.copy(), p0=p0, xend=xnew, Unew=Unew, Hcur=Hcur,
                                   Hnew=Hnew, u=u, ok=ok, accept=accept, xres=xres.copy(), Ures=Ures.copy()))
        return xres, Ures, dres, accept

    @with_host_threads
    def sample_flow(self, x_init=None, pipeline=True, max_steps=None, step_hook=None, device_restart=True, async_handback=True,
                    resume=False):
        """Same chains, same samples as sample() (each chain consumes its own RNG stream in the reference's order and
        chains never interact), scheduled as a continuous flow: every device step evaluates every chain once, each
        chain at its own point of its own trajectory (rfs_flow_step), and a chain that finishes a trajectory is
        accepted / rejected and restarted on the spot instead of waiting for the longest trajectory of the batch.
        With L drawn per chain in [Lmin, Lmax] the batch schedule spends Lmax + 1 evaluations per round and chain,
        this one mean(L) + 2.
        ``checkpoint`` / ``checkpoint_every`` / ``mass_adapt`` (constructor) work on trajectory COUNTS, as in sample(): the
        flow runs in segments that end when every chain has completed that many trajectories (a chain that gets there early
        idles until the last one has: a bubble of about one trajectory per segment).  At such a barrier the state is the one
        sample() checkpoints -- models, counters, sample slots, every chain's RNG position -- so the same file format serves
        both schedules, ``resume=True`` continues from ``self.checkpoint`` with the samples of an uninterrupted run, and the
        mass matrix is re-estimated from the ensemble exactly where sample() does it.  Where the write policy differs from
        sample(): a checkpoint is written only at a barrier with chains still unfinished (sample() also writes one whenever it
        stops early); `checkpoint` without `checkpoint_every` is refused."""
        import torch
        nc, ns, nd_ = self.nchains, self.nsamples, self.ndraws
        if self.checkpoint and not self.checkpoint_every and not resume:
            # sample() writes a checkpoint whenever it stops unfinished; the flow has states a run can continue from only at its
            # barriers (every chain at the same trajectory count), so without an interval there would never be a file
            raise ValueError("sample_flow: `checkpoint` needs `checkpoint_every` > 0 (checkpoints are written at barriers: every "
                             "chain at the same trajectory count; a run cut by max_steps inside a segment writes none)")
        if self.inverse_mass is not None:
            self.model.set_inverse_mass(self.inverse_mass)
        dev = self._device()
        ndata = self.model.dobs.shape[0]
        total = nd_ + ns
        if resume:
            ck = load_checkpoint(self.checkpoint, self.rng)
            x, i, ncount = ck["x"], ck["i"], ck["ncount"]
            misfit, x_cache, self.initmodel = ck["misfit"], ck["x_cache"], ck["initmodel"]
            syndata = ck["syndata"] if "syndata" in ck else None
            self.ii = int(ck["ii"])
            cur = int(ck["ntraj"]) if "ntraj" in ck else 0
            if "inverse_mass" in ck:
                self._set_inverse_mass(ck["inverse_mass"])
            nx = x.shape[1]
        else:
            x = initial_models(self.rng, self.boundaries) if x_init is None else np.array(x_init, dtype=float)
            self.initmodel = x.copy()
            nx = x.shape[1]
            misfit = np.zeros((nc, ns)); x_cache = np.zeros((nc, ns, nx))
            syndata = np.zeros((nc, ns, ndata)) if self.store_syn else None
            i = np.zeros(nc, dtype=int); ncount = np.zeros(nc, dtype=int)
            cur = 0
        self.live_counts = (i, ncount)          # accepted / completed trajectories per chain, as the books stand (step hooks read them)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        sampler = self
        self.flow_withdrawn = 0                     # early draws taken back because the trajectory failed after all
        self.flow_steps = 0
        every = self.checkpoint_every if self.checkpoint else 0
        points = sorted(self.mass_adapt) if self.mass_adapt else []

        def next_barrier(c):
            """Smallest trajectory count > c at which every chain has to stand still (checkpoint, mass adaptation)."""
            cand = [p for p in points if p > c]
            if every:
                cand.append((c // every + 1) * every)
            return min(cand) if cand else None

        def book(idx, accept, Unew, xend, dnew):
            """One completed trajectory per chain of idx: keep the accepted end points, fill the sample slots."""
            ca = idx[accept]
            if len(ca):
                x[ca] = xend[accept]
                keep = i[ca] >= nd_
                if np.any(keep):
                    ck_, slot = ca[keep], i[ca][keep] - nd_
                    misfit[ck_, slot] = Unew[accept][keep]
                    x_cache[ck_, slot] = xend[accept][keep]
                    if syndata is not None:
                        syndata[ck_, slot] = dnew[accept][keep]
                i[ca] += 1; self.ii += len(ca)
            ncount[idx] += 1

        capped = False
        while np.any(i < total) and not capped:
            if cur in points:                       # as sample(): the ensemble's variance, before trajectory number `cur`
                self._set_inverse_mass(ensemble_inverse_mass(x))
            target = next_barrier(cur)              # None: no barrier ahead, the segment runs to the end
            lim = np.inf if target is None else target
            live = np.nonzero(i < total)[0]
            st = self.model.flow_state(t(x), torch.full((nc,), float(self.dt), dtype=torch.float64, device=dev),
                                       t(self.boundaries))
            livel = [int(c) for c in live]
            L = np.zeros(nc, dtype=np.int64)
            L[live] = self.rng.randint(livel, self.Lrange[0], self.Lrange[1] + 1)  # hmc.py:248, then :146
            p0 = np.zeros((nc, nx)); p0[live] = self.rng.randn(livel, nx) * self._pscale
            st["p"].copy_(t(p0))
            rem0 = np.where(i < total, L, -1).astype(np.int32)
            st["rem"].copy_(t(rem0)); st["fresh"].copy_(t((i < total).astype(np.int32)))
            pending = {}                            # chain -> snapshot its early draws can be undone with

            def process_done(idx, res):
                ok = res["ok"].astype(bool)
                Hcur, Hnew, Unew, Ucur, xend = res["Hcur"], res["Hnew"], res["Unew"], res["Ucur"], res["x"]
                dnew = res.get("dsyn_new")
                u = np.full(len(idx), np.nan)
                u[ok] = self.rng.rand([int(c) for c in idx[ok]])                    # hmc.py:193, skipped on failure
                with np.errstate(over="ignore", invalid="ignore"):
                    accept = ok & (u < np.exp(-(Hnew - Hcur)))
                book(idx, accept, Unew, xend, dnew)
                restart = [int(c) for c in idx[(i[idx] < total) & (ncount[idx] < lim)]]
                if self.verbose:
                    for k, c in enumerate(idx):
                        if i[c] % 50 == 0 or i[c] == ns - 1:
                            Uc = Unew[k] if accept[k] else (Ucur[k] if ok[k] else np.inf)
                            print("chain {}: {:.2%}, misfit={:.3} -- accept ratio {:.2%}".format(
                                self.first_chain + c, i[c] / total, Uc, i[c] / ncount[c]))
                    sys.stdout.flush()
                # every finished chain goes back to the model it keeps (accepted end point or its start model) and the
                # unfinished ones restart: randint(L) then randn(p0), the reference's draw order (hmc.py:248, :146)
                rs = None
                if restart:
                    Lr = self.rng.randint(restart, self.Lrange[0], self.Lrange[1] + 1)
                    rs = dict(idx=restart, p=self.rng.randn(restart, nx) * self._pscale, rem=Lr)
                return x[idx], rs

            # Restarts on the device (rfs_flow_step2): neither the acceptance draw nor the next L and momentum depend on the
            # trajectory (hmc.py:193, 248, 146), so they are drawn -- in that order, from the chain's own stream -- while the
            # trajectory still runs, and the device accepts / rejects and starts over by itself.  The one exception is the
            # reference's failure paths, which skip the acceptance draw (hmc.py:156,173,177,179): the streams are
            # snapshotted before the early draws and rewound for a chain that fails.
            class Restart:
                rem0 = np.where(i < total, L, 1 << 30)      # (idle chains: never "about to complete")

                @staticmethod
                def predraw(cands):
                    # needs another trajectory whatever the decision, and on this side of the barrier
                    sel = cands[(i[cands] + 1 < total) & (ncount[cands] + 1 < lim)]
                    if len(sel) == 0:
                        return sel, None, None, None
                    snap = sampler.rng.snapshot(sel)
                    for c in sel:
                        pending[int(c)] = snap
                    cl = [int(c) for c in sel]
                    u = sampler.rng.rand(cl)
                    Ln = sampler.rng.randint(cl, sampler.Lrange[0], sampler.Lrange[1] + 1)
                    return sel, u, sampler.rng.randn(cl, nx) * sampler._pscale, Ln

                @staticmethod
                def done(idx, res, accepted):
                    if sampler.trace is not None:       # (diagnostics: one record per batch of trajectories the device completed)
                        sampler.trace.append(dict(active=[int(c) for c in idx], Hcur=res["Hcur"].copy(), Hnew=res["Hnew"].copy(),
                                                  Unew=res["Unew"].copy(), accept=np.asarray(accepted).copy()))
                    book(idx, accepted, res["Unew"], res["x"], res.get("dsyn_new"))
                    for c in idx:
                        pending.pop(int(c), None)

                @staticmethod
                def withdraw(idx):
                    sampler.flow_withdrawn += len(idx)
                    for c in idx:
                        sampler.rng.restore(pending.pop(int(c)), [int(c)])

            left = None if max_steps is None else max_steps - self.flow_steps
            base = self.flow_steps
            hook = None if step_hook is None else (lambda s_, st_, _b=base: step_hook(_b + s_, st_))
            nst = run_flow(self.model, st, process_done, lambda: bool(np.any((i < total) & (ncount < lim))),
                           fetch_syn=syndata is not None, pipeline=pipeline, max_steps=left,
                           step_hook=hook, restart=Restart if device_restart else None,
                           async_handback=async_handback)
            self.flow_steps += nst
            capped = max_steps is not None and self.flow_steps >= max_steps
            if capped and np.any((i < total) & (ncount < lim)):
                break                                   # stopped inside a segment: no barrier state to keep
            if target is not None:
                cur = target
                if every and cur % every == 0 and np.any(i < total):
                    self._save_checkpoint(x, np.zeros(nc), i, ncount, misfit, x_cache, syndata, cur)
            else:
                break
        self.finished = not bool(np.any(i < total))
        self.naccepted, self.ntrajectories = i.copy(), ncount.copy()
        if not self.finished:                    # stopped by max_steps: nothing is written
            self.x_cache = x_cache
            return misfit[0] if nc == 1 else misfit
        return self._finish(misfit, x_cache, syndata, i, ncount)

    @with_host_threads
    def sample(self, x_init=None, resume=False, max_trajectories=None):
        """pyhmc/hmc.py:228-276.  Returns misfit[nsamples] (nchains == 1) or [nchains, nsamples].
        ``resume``: continue from ``self.checkpoint`` (same results as an uninterrupted run);
        ``max_trajectories``: stop after that many outer iterations (the checkpoint is written first)."""
        nc, ns, nd_ = self.nchains, self.nsamples, self.ndraws
        if self.inverse_mass is not None:
            self.model.set_inverse_mass(self.inverse_mass)
        ndata = self.model.dobs.shape[0]
        total = nd_ + ns
        if resume:
            st = load_checkpoint(self.checkpoint, self.rng)
            x, U, i, ncount = st["x"], st["U"], st["i"], st["ncount"]
            misfit, x_cache, self.initmodel = st["misfit"], st["x_cache"], st["initmodel"]
            syndata = st["syndata"] if "syndata" in st else None
            self.ii = int(st["ii"])
            nx = x.shape[1]
            ntraj = int(st["ntraj"]) if "ntraj" in st else 0
            if "inverse_mass" in st:
                self._set_inverse_mass(st["inverse_mass"])
        else:
            x = initial_models(self.rng, self.boundaries) if x_init is None else np.array(x_init, dtype=float)
            self.initmodel = x.copy()
            nx = x.shape[1]
            misfit = np.zeros((nc, ns))
            x_cache = np.zeros((nc, ns, nx))
            syndata = np.zeros((nc, ns, ndata)) if self.store_syn else None
            i = np.zeros(nc, dtype=int)
            ncount = np.zeros(nc, dtype=int)
            U = np.zeros(nc)
            ntraj = 0
        ntraj0 = ntraj
        while np.any(i < total):
            if max_trajectories is not None and ntraj - ntraj0 >= max_trajectories:
                break
            if self.mass_adapt and ntraj in self.mass_adapt:
                self._set_inverse_mass(ensemble_inverse_mass(x))
            active = [c for c in range(nc) if i[c] < total]
            L = self.rng.randint(active, self.Lrange[0], self.Lrange[1] + 1)  # hmc.py:248
            xa, Ua, da, acc = self._leapfrog(x, active, L)
            for k, c in enumerate(active):
                x[c] = xa[k]; U[c] = Ua[k]
                if acc[k]:
                    if i[c] >= nd_:
                        misfit[c, i[c] - nd_] = Ua[k]
                        x_cache[c, i[c] - nd_] = xa[k]
                        if syndata is not None:
                            syndata[c, i[c] - nd_] = da[k]
                    i[c] += 1
                    self.ii += 1
                ncount[c] += 1
                if self.verbose and (i[c] % 50 == 0 or i[c] == ns - 1):
                    print("chain {}: {:.2%}, misfit={:.3} -- accept ratio {:.2%}".format(
                        self.first_chain + c, i[c] / total, U[c], i[c] / ncount[c]))
                    sys.stdout.flush()
            ntraj += 1
            if self.checkpoint and self.checkpoint_every and ntraj % self.checkpoint_every == 0:
                self._save_checkpoint(x, U, i, ncount, misfit, x_cache, syndata, ntraj)
        if self.checkpoint and np.any(i < total):
            self._save_checkpoint(x, U, i, ncount, misfit, x_cache, syndata, ntraj)
            self.finished = False
            return misfit[0] if nc == 1 else misfit
        self.finished = True
        return self._finish(misfit, x_cache, syndata, i, ncount)

    def _finish(self, misfit, x_cache, syndata, i, ncount):
        nc, nx = self.nchains, x_cache.shape[2]
        self.accept_ratio = i / np.maximum(ncount, 1)
        # mean of the nbest lowest-misfit samples, one more evaluation (hmc.py:266-275)
        xmean = np.zeros((nc, nx))
        for c in range(nc):
            idx = np.argsort(misfit[c])
            xmean[c] = np.mean(x_cache[c, idx[:self.nbest_model]], axis=0)
        res = self.model.misfit_and_grad(xmean)
        synmean = res[2]
        self.x_cache, self.syndata, self.xmean, self.synmean = x_cache, syndata, xmean, synmean
        if self.outdir is not None:
            fmt = resolve_store_format(self.store_format)
            self.result_file = save_batched_results(self.outdir, self.name, self.myrank, self.first_chain,
                                                    self.initmodel, self.model.dobs, xmean, synmean, x_cache,
                                                    syndata, misfit, fmt=fmt)
            if self.per_chain_files:
                for c in range(nc):
                    save_chain_results(self.outdir, self.name, self.first_chain + c, self.initmodel[c],
                                       self.model.dobs, xmean[c], synmean[c], x_cache[c],
                                       None if syndata is None else syndata[c], fmt=fmt)
        return misfit[0] if nc == 1 else misfit

    def _save_checkpoint(self, x, U, i, ncount, misfit, x_cache, syndata, ntraj):
        if hasattr(self.model, "reset_warm_start"):      # the run may be cut here: what follows starts from the full search, as a resumed run would
            self.model.reset_warm_start()
        save_checkpoint(self.checkpoint, self.rng, x=x, U=U, i=i, ncount=ncount, misfit=misfit, x_cache=x_cache,
                        syndata=syndata, initmodel=self.initmodel, ii=self.ii, ntraj=ntraj,
                        inverse_mass=self.inverse_mass)
