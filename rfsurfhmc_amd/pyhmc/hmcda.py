"""HMCDualAveraging -- batched mirror of the reference sampler pyhmc/hmcda.py (HMC with
dual-averaging step-size adaptation).  Per-chain dt and L = max(1, int(lambda/dt)) are handled by
the device trajectory (rfs_leapfrog_dev takes dt[chain], L[chain]).  Draw order per chain and
iteration as in the reference: randn(n)*0.5, [trajectory], rand() -- the rand() is always drawn
(hmcda.py:311)."""
import sys

import numpy as np

from ._batched import (ChainRNG, ensemble_inverse_mass, initial_models, load_checkpoint, run_flow, save_batched_results,
                       save_chain_results, save_checkpoint)
from ._batched import store_format as resolve_store_format, with_host_threads


def _mirror(x, p, boundaries):
    """pyhmc/hmcda.py:152-168 on [nchain, n] arrays.  64 reflections as the reference makes them; a point that is still outside
    then (a momentum that has blown up) is folded in closed form -- the same point up to rounding, where the reference would loop
    on, for ever if the point is infinite -- and a non-finite one is put in the middle of its bounds (its energy is not finite:
    the trajectory is rejected).  Same rule as the device's flow_mirror (csrc/rfsurf_kernels.hpp)."""
    x, p = x.copy(), p.copy()
    high, low = boundaries[:, 1][None, :], boundaries[:, 0][None, :]
    idx1, idx2 = x > high, x < low
    it = 0
    while np.any(idx1 | idx2) and it < 64:
        x = np.where(idx1, 2 * high - x, x); p = np.where(idx1, -p, p)
        idx2 = x < low
        x = np.where(idx2, 2 * low - x, x); p = np.where(idx2, -p, p)
        idx1, idx2 = x > high, x < low
        it += 1
    out = (x > high) | (x < low) | ~np.isfinite(x)
    if np.any(out):
        w = np.broadcast_to(high - low, x.shape)
        lo = np.broadcast_to(low, x.shape)
        fold = out & np.isfinite(x) & (np.abs(x) < 1.0e300) & (w > 0)
        with np.errstate(invalid="ignore"):
            y = np.mod(x - lo, 2.0 * w)                      # (numpy's mod: in [0, 2 w) for a positive divisor)
        odd = y > w
        x = np.where(fold, lo + np.where(odd, 2.0 * w - y, y), x)
        p = np.where(fold & odd, -p, p)
        x = np.where(out & ~fold, lo + 0.5 * w, x)
    return x, p


class HMCDualAveraging:
    def __init__(self, UserDefinedModel, boundaries, dt, L0, nbest_model, target_ratio, seed, nsamples, ndraws,
                 myrank=0, name="mychain", outdir="./", nchains=1, store_syn=True, verbose=True,
                 per_chain_files=None, checkpoint=None, checkpoint_every=0, inverse_mass=None, mass_adapt=None,
                 L_cap=None, store_format="auto", warm_start=None):
        self.model = UserDefinedModel
        self.boundaries = np.asarray(boundaries, dtype=np.float64)
        self.dt, self.L = dt, L0
        # Longest trajectory a chain may ask for.  The reference's L = max(1, int(lambda / dt)) (hmcda.py:307) is
        # unbounded: a chain whose dt collapses during burn-in only slows its own MPI rank there, but in a batch it
        # would stall every chain of the rank (and overflow int32).  Such a chain is clamped to L_cap steps (and
        # reported once); chains that stay below the cap are untouched.  Default: 100 L0.
        self.L_cap = int(L_cap) if L_cap is not None else max(1, 100 * int(L0))
        self._cap_warned = False
        self.nbest_model, self.nsamples, self.ndraws = nbest_model, nsamples, ndraws
        if ndraws < 0.1 * nsamples:                                           # hmcda.py:57-60
            raise ValueError(f"in dual averaging, ndraws should > nsamples * 0.1 (ndraws = {ndraws}, nsamples = {nsamples})")
        self.myrank, self.nchains = myrank, int(nchains)
        self.first_chain = myrank * self.nchains
        self.seed, self.name, self.outdir = seed, name, outdir
        self.store_syn, self.verbose = store_syn, verbose
        # result files: "h5" = the reference's HDF5 (pyhmc/hmc.py:58,203-226), "npz", or "auto" = HDF5 where h5py or
        # libhdf5 is present; resolved when the files are written so that a missing library cannot fail a finished run early
        self.store_format = store_format
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
        self.delta = target_ratio                                             # hmcda.py:70-76
        self._h0, self._gamma, self._t0, self._kappa = 0.0, 0.05, 10.0, 0.75
        self._lambda = L0 * self.dt
        self.rng = ChainRNG(seed, self.first_chain, self.nchains)
        self.ii = 0
        self.trace = None

    @classmethod
    def init(cls, UserDefinedModel, boundaries, rank, **kargs):
        """pyhmc/hmcda.py:84-97 (+ optional keys ``nchains``, ``mass_adapt``)."""
        return cls(UserDefinedModel, boundaries, kargs["dt"], kargs["L0"], kargs["nbest"], kargs["target_ratio"],
                   kargs["seed"], kargs["nsamples"], kargs["ndraws"], rank, kargs["name"], kargs["OUTPUT_DIR"],
                   nchains=kargs.get("nchains", 1), mass_adapt=kargs.get("mass_adapt"),
                   checkpoint=kargs.get("checkpoint"), checkpoint_every=kargs.get("checkpoint_every", 0),
                   L_cap=kargs.get("L_cap"), store_format=kargs.get("store_format", "auto"), warm_start=kargs.get("warm_start"))

    def _traj_len(self, dt):
        """L = max(1, int(lambda / dt)) per chain (hmcda.py:307), clamped to L_cap before the integer cast."""
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            lf = np.floor(self._lambda / np.asarray(dt, dtype=np.float64))
        lf = np.where(np.isfinite(lf), lf, float(self.L_cap))
        # a run in which this count is not zero departed from the reference's unbounded L; it is kept with the results
        self.n_L_clamped = getattr(self, "n_L_clamped", 0) + int(np.sum(lf > self.L_cap))
        if not self._cap_warned and np.any(lf > self.L_cap):
            self._cap_warned = True
            print(f"HMCDualAveraging: {int(np.sum(lf > self.L_cap))} chain(s) ask for more than L_cap = {self.L_cap} "
                  "leapfrog steps (collapsed dt); clamped", file=sys.stderr)
        return np.clip(lf, 1.0, float(self.L_cap)).astype(np.int32)

    def _set_inverse_mass(self, minv):
        self.inverse_mass = np.asarray(minv, dtype=np.float64)
        self._pscale = 0.5 / np.sqrt(self.inverse_mass)
        self.model.set_inverse_mass(self.inverse_mass)

    def _device(self):
        import torch
        dev = getattr(self.model, "torch_device", None)       # host-logic tests plug in a CPU model here
        return dev if dev is not None else torch.device("cuda", getattr(self.model, "device", 0) or 0)

    def _find_initial_dt(self, dt0, x):
        """pyhmc/hmcda.py:170-220, all chains at once (masked): in effect dt is doubled while the
        one-step acceptance stays above 0.5, at most 20 times."""
        nc, n = x.shape
        idx = list(range(nc))
        xcur = x.copy()
        dt = np.full(nc, float(dt0))
        pcur = self.rng.randn(idx, n) * self._pscale
        mi = 1.0 if self.inverse_mass is None else self.inverse_mass[None, :]      # K = p.M^-1 p / 2, x' = M^-1 p
        U, grad, _, flag = self.model.misfit_and_grad(xcur)
        Hcur = U + 0.5 * np.sum(pcur * pcur * mi, axis=1)
        a = np.zeros(nc)
        # A chain whose evaluation fails here, or comes back with a NaN gradient (the reference's sregn96 where a root equals a
        # layer velocity: about one chain in 8192 per evaluation), keeps dt0.  The reference -- one chain per process -- exits on
        # the failed flag (:196-198) and would carry a NaN gradient into its next model; a batch does not give up 8191 chains
        # for one, and says so.
        live = flag & np.isfinite(U) & np.isfinite(grad).all(axis=1)
        nbad = int((~live).sum())
        grad = np.where(live[:, None], grad, 0.0)
        pcur = pcur - 0.5 * dt[:, None] * grad
        for it in range(20):
            xn, pn = _mirror(xcur + dt[:, None] * (pcur * mi), pcur, self.boundaries)
            xcur = np.where(live[:, None], xn, xcur); pcur = np.where(live[:, None], pn, pcur)
            U, grad, _, flag = self.model.misfit_and_grad(xcur)
            failed = live & ~flag
            if np.any(failed):
                nbad += int(failed.sum()); live = live & flag
            grad = np.where(live[:, None], grad, 0.0)
            pcur = np.where(live[:, None], pcur - 0.5 * dt[:, None] * grad, pcur)
            Hnew = U + 0.5 * np.sum(pcur * pcur * mi, axis=1)
            ediff = -(Hnew - Hcur)
            if it == 0:
                a = 2.0 * (ediff > np.log(0.5)) - 1.0
            # (a non-finite energy -- the reference's sregn96 returns NaN kernels where a root equals a layer velocity -- never
            # satisfies the reference's test (:209) and its loop would carry NaN models into the next evaluation; such a
            # chain stops here and keeps the step size it has)
            stop = live & ((ediff < np.log(0.5)) | ~np.isfinite(ediff))
            go = live & ~stop
            pcur = np.where(go[:, None], pcur - 0.5 * dt[:, None] * grad, pcur)
            Hcur = np.where(go, Hnew, Hcur)
            dt = np.where(go, dt * 2.0 ** a, dt)
            live = go
            if not live.any():
                break
        if nbad:
            print(f"HMCDualAveraging._find_initial_dt: {nbad} chain(s) with a failed or non-finite evaluation keep their step size "
                  f"(the reference exits there: 'error in chain')", file=sys.stderr)
        if self.verbose:
            for c in range(nc):
                print(f"chain {self.first_chain + c}: change dt from {dt0} to {dt[c]}")
        return dt

    def _leapfrog(self, x, dt, L):
        """pyhmc/hmcda.py:222-278 for all chains: returns (xnew, Unew, dsyn_new, alpha)."""
        import torch
        dev = self._device()
        nc, n = x.shape
        idx = list(range(nc))
        p0 = self.rng.randn(idx, n) * self._pscale
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        out = self.model.leapfrog_device(t(x), t(p0), t(dt.astype(np.float64)), t(L.astype(np.int32)), t(self.boundaries))
        ok = out["ok"].cpu().numpy().astype(bool)
        Hcur, Hnew = out["Hcur"].cpu().numpy(), out["Hnew"].cpu().numpy()
        with np.errstate(over="ignore", invalid="ignore"):
            alpha = np.where(ok, np.minimum(1.0, np.exp(-(Hnew - Hcur))), 0.0)
        xnew = np.where(ok[:, None], out["xnew"].cpu().numpy(), x)
        Unew = np.where(ok, out["Unew"].cpu().numpy(), np.inf)
        dnew = np.where(ok[:, None], out["dsyn_new"].cpu().numpy(), self.model.dobs[None, :])
        if self.trace is not None:
            self.trace.append(dict(L=L.copy(), dt=dt.copy(), p0=p0, xend=xnew.copy(), Unew=Unew.copy(), Hcur=Hcur,
                                   Hnew=Hnew, alpha=alpha.copy(), ok=ok))
        return xnew, Unew, dnew, alpha

    @with_host_threads
    def sample(self, x_init=None, resume=False, max_trajectories=None):
        """pyhmc/hmcda.py:280-369.  ``resume`` / ``max_trajectories``: see HamitonianMC.sample."""
        nc, ns, nd_ = self.nchains, self.nsamples, self.ndraws
        if self.inverse_mass is not None:
            self.model.set_inverse_mass(self.inverse_mass)
        ndata = self.model.dobs.shape[0]
        mu = np.log(10 * self.dt)
        total = nd_ + ns
        if resume:
            st = load_checkpoint(self.checkpoint, self.rng)
            x, i, ncount = st["x"], st["i"], st["ncount"]
            misfit, x_cache, self.initmodel = st["misfit"], st["x_cache"], st["initmodel"]
            syndata = st["syndata"] if "syndata" in st else None
            dt, dtbar, h0 = st["dt"], st["dtbar"], st["h0"]
            self.ii = int(st["ii"])
            nx = x.shape[1]
            ntraj = int(st["ntraj"]) if "ntraj" in st else 0
            if "inverse_mass" in st:
                self._set_inverse_mass(st["inverse_mass"])
        else:
            x = initial_models(self.rng, self.boundaries) if x_init is None else np.array(x_init, dtype=float)
            self.initmodel = x.copy()
            nx = x.shape[1]
            misfit = np.zeros((nc, ns)); x_cache = np.zeros((nc, ns, nx))
            syndata = np.zeros((nc, ns, ndata)) if self.store_syn else None
            dt = self._find_initial_dt(self.dt, x)
            dtbar = dt * 1.0
            h0 = np.full(nc, self._h0)
            i = np.zeros(nc, dtype=int); ncount = np.zeros(nc, dtype=int)
            ntraj = 0
        ntraj0 = ntraj
        idx_all = list(range(nc))
        U = np.zeros(nc)
        while np.any(i < total):
            if max_trajectories is not None and ntraj - ntraj0 >= max_trajectories:
                break
            if self.mass_adapt and ntraj in self.mass_adapt:        # dual averaging then re-tunes dt (burn-in)
                self._set_inverse_mass(ensemble_inverse_mass(x))
            live = i < total
            L = self._traj_len(dt)                                                  # hmcda.py:307
            x1, U, dsyn, alpha = self._leapfrog(x, dt, L)
            u = self.rng.rand(idx_all)
            acc = live & (u < alpha)
            for c in np.nonzero(acc)[0]:
                x[c] = x1[c]
                if i[c] >= nd_:
                    misfit[c, i[c] - nd_] = U[c]; x_cache[c, i[c] - nd_] = x1[c]
                    if syndata is not None:
                        syndata[c, i[c] - nd_] = dsyn[c]
                i[c] += 1; self.ii += 1
            # dual averaging (hmcda.py:329-345)
            adapt = live & (ncount < nd_)
            m = ncount + 1.0
            fac = 1.0 / (m + self._t0)
            h_new = (1 - fac) * h0 + fac * (self.delta - alpha)
            logdt = mu - np.sqrt(m) / self._gamma * h_new
            fac2 = m ** (-self._kappa)
            dtbar_new = np.exp(fac2 * logdt + (1 - fac2) * np.log(dtbar))
            h0 = np.where(adapt, h_new, h0)
            dt = np.where(adapt, np.exp(logdt), np.where(live, dtbar, dt))
            dtbar = np.where(adapt, dtbar_new, dtbar)
            ncount = ncount + live
            if self.verbose:
                for c in np.nonzero(live)[0]:
                    if i[c] % 50 == 0 or i[c] == ns - 1:
                        print("chain {}: {:.2%}, dt = {:.3},  misfit={:.3} -- accept ratio {:.2%}".format(
                            self.first_chain + c, i[c] / total, dt[c], U[c], i[c] / ncount[c]))
                sys.stdout.flush()
            ntraj += 1
            if self.checkpoint and self.checkpoint_every and ntraj % self.checkpoint_every == 0:
                self._save_checkpoint(x, i, ncount, misfit, x_cache, syndata, dt, dtbar, h0, ntraj)
        if self.checkpoint and np.any(i < total):
            self._save_checkpoint(x, i, ncount, misfit, x_cache, syndata, dt, dtbar, h0, ntraj)
            self.finished = False
            return misfit[0] if nc == 1 else misfit
        self.finished = True
        return self._finish(misfit, x_cache, syndata, i, ncount, dt)

    def _finish(self, misfit, x_cache, syndata, i, ncount, dt):
        nc, nx = self.nchains, x_cache.shape[2]
        self.dt_final, self.accept_ratio = dt, i / np.maximum(ncount, 1)
        nbests = 10                                                           # hard-coded, hmcda.py:359
        xmean = np.zeros((nc, nx))
        for c in range(nc):
            idx = np.argsort(misfit[c])
            xmean[c] = np.mean(x_cache[c, idx[:nbests]], axis=0)
        synmean = self.model.misfit_and_grad(xmean)[2]
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

    @with_host_threads
    def sample_flow(self, x_init=None, pipeline=True, max_steps=None, step_hook=None, device_restart=True, async_handback=True,
                    resume=False):
        """Same chains and samples as sample(), on the continuous-flow schedule (rfs_flow_step): with dual averaging
        every chain has its own step size and therefore its own trajectory length L = max(1, int(lambda / dt))
        (hmcda.py:307); here no chain waits for the longest one.  Per chain the RNG stream is consumed in the reference's
        order (momentum at the start of a trajectory, the acceptance draw at its end).
        ``checkpoint`` / ``checkpoint_every`` / ``mass_adapt`` / ``resume``: as HamitonianMC.sample_flow -- segments that end
        at trajectory counts, sample()'s checkpoint format (with dt, dtbar and the dual-averaging statistic per chain)."""
        import torch
        nc, ns, nd_ = self.nchains, self.nsamples, self.ndraws
        if self.checkpoint and not self.checkpoint_every and not resume:      # (as HamitonianMC.sample_flow)
            raise ValueError("sample_flow: `checkpoint` needs `checkpoint_every` > 0 (checkpoints are written at barriers)")
        if self.inverse_mass is not None:
            self.model.set_inverse_mass(self.inverse_mass)
        dev = self._device()
        ndata = self.model.dobs.shape[0]
        mu = np.log(10 * self.dt)
        total = nd_ + ns
        if resume:
            ck = load_checkpoint(self.checkpoint, self.rng)
            x, i, ncount = ck["x"], ck["i"], ck["ncount"]
            misfit, x_cache, self.initmodel = ck["misfit"], ck["x_cache"], ck["initmodel"]
            syndata = ck["syndata"] if "syndata" in ck else None
            dt, dtbar, h0 = ck["dt"], ck["dtbar"], ck["h0"]
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
            dt = self._find_initial_dt(self.dt, x)
            dtbar = dt * 1.0
            h0 = np.full(nc, self._h0)
            i = np.zeros(nc, dtype=int); ncount = np.zeros(nc, dtype=int)
            cur = 0
        self.live_counts = (i, ncount)          # accepted / completed trajectories per chain, as the books stand (step hooks read them)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        sampler = self
        self.flow_steps = 0
        every = self.checkpoint_every if self.checkpoint else 0
        points = sorted(self.mass_adapt) if self.mass_adapt else []

        def next_barrier(c):
            cand = [p for p in points if p > c]
            if every:
                cand.append((c // every + 1) * every)
            return min(cand) if cand else None

        capped = False
        while np.any(i < total) and not capped:
          if cur in points:                         # as sample(): dual averaging then re-tunes dt (burn-in)
              self._set_inverse_mass(ensemble_inverse_mass(x))
          target = next_barrier(cur)
          lim = np.inf if target is None else target
          live0 = i < total
          st = self.model.flow_state(t(x), t(dt.astype(np.float64)), t(self.boundaries))
          livel = [int(c) for c in np.nonzero(live0)[0]]
          p0 = np.zeros((nc, nx)); p0[live0] = self.rng.randn(livel, nx) * self._pscale
          st["p"].copy_(t(p0))
          # (one call per segment, for the live chains only: _traj_len counts the lengths it clamps to L_cap, and that count is
          # part of the results -- ADVICE r04)
          L0 = np.zeros(nc, dtype=np.int64); L0[live0] = self._traj_len(dt[live0])
          st["rem"].copy_(t(np.where(live0, L0, -1).astype(np.int32)))
          st["fresh"].copy_(t(live0.astype(np.int32)))
          pending = {}                                # chain -> (u, p) drawn ahead of time for it (restarts on the device)

          def books(idx, ok, Hcur, Hnew, Unew, xend, dnew, u, acc=None):
              """One completed trajectory per chain of idx: accept / reject (acc given: the device's decision with the same
              u), sample slots, dual averaging of the step size (hmcda.py:329-345)."""
              Unew = np.where(ok, Unew, np.inf)
              with np.errstate(over="ignore", invalid="ignore"):
                  alpha = np.where(ok, np.minimum(1.0, np.exp(-(Hnew - Hcur))), 0.0)
              if acc is None:
                  acc = u < alpha
              # accepted end points (vectorised over the finished chains; one sample slot per chain and trajectory)
              ca = idx[acc]
              if len(ca):
                  x[ca] = xend[acc]
                  keep = i[ca] >= nd_
                  if np.any(keep):
                      ck, slot = ca[keep], i[ca][keep] - nd_
                      misfit[ck, slot] = Unew[acc][keep]
                      x_cache[ck, slot] = xend[acc][keep]
                      if syndata is not None:
                          okk = ok[acc][keep]
                          syndata[ck, slot] = np.where(okk[:, None], dnew[acc][keep], self.model.dobs[None, :])
                  i[ca] += 1; self.ii += len(ca)
              adapt = ncount[idx] < nd_
              m = ncount[idx] + 1.0
              fac = 1.0 / (m + self._t0)
              h_new = (1 - fac) * h0[idx] + fac * (self.delta - alpha)
              logdt = mu - np.sqrt(m) / self._gamma * h_new
              fac2 = m ** (-self._kappa)
              dtbar_new = np.exp(fac2 * logdt + (1 - fac2) * np.log(dtbar[idx]))
              h0[idx] = np.where(adapt, h_new, h0[idx])
              dt[idx] = np.where(adapt, np.exp(logdt), dtbar[idx])
              dtbar[idx] = np.where(adapt, dtbar_new, dtbar[idx])
              ncount[idx] += 1
              if self.verbose:
                  for k, c in enumerate(idx):
                      if i[c] % 50 == 0 or i[c] == ns - 1:
                          print("chain {}: {:.2%}, dt = {:.3},  misfit={:.3} -- accept ratio {:.2%}".format(
                              self.first_chain + c, i[c] / total, dt[c], Unew[k], i[c] / ncount[c]))
                  sys.stdout.flush()

          def process_done(idx, res):
              ok = res["ok"].astype(bool)
              # acceptance draws (every iteration, hmcda.py:311): a chain that had its draws made early but failed -- the
              # device leaves those to the host -- uses exactly those, the others draw now
              early = np.array([int(c) in pending for c in idx], dtype=bool)
              u = np.empty(len(idx))
              if early.any():
                  u[early] = [pending[int(c)][0] for c in idx[early]]
              if (~early).any():
                  u[~early] = self.rng.rand([int(c) for c in idx[~early]])
              books(idx, ok, res["Hcur"], res["Hnew"], res["Unew"], res["x"], res.get("dsyn_new"), u)
              restart = [int(c) for c in idx if i[c] < total and ncount[c] < lim]
              rs = None
              if restart:
                  fresh_p = [c for c in restart if c not in pending]
                  drawn = dict(zip(fresh_p, self.rng.randn(fresh_p, nx) * self._pscale)) if fresh_p else {}
                  pr = np.stack([pending[c][1] if c in pending else drawn[c] for c in restart])
                  rs = dict(idx=restart, p=pr, dt=dt[restart], rem=self._traj_len(dt[restart]))
              for c in idx:
                  pending.pop(int(c), None)
              return x[idx], rs

          # Restarts on the device, deferred form (rfs_flow_step2 with gsave / kick): the acceptance draw and the next
          # momentum never depend on the trajectory (hmcda.py:311, :236) and are drawn one step ahead; the next step size
          # does (dual averaging), so the device accepts / rejects, starts the chain on its new momentum and evaluates the
          # start model at once, while the first half kick waits one call for dt and L from the host.
          sampler = self

          class Restart:
              rem0 = np.where(live0, L0, 1 << 30)
              deferred = True

              @staticmethod
              def predraw(cands):
                  sel = cands[(i[cands] + 1 < total) & (ncount[cands] + 1 < lim)]
                  if len(sel) == 0:
                      return sel, None, None, None
                  cl = [int(c) for c in sel]
                  u = sampler.rng.rand(cl)
                  pn = sampler.rng.randn(cl, nx) * sampler._pscale
                  for k, c in enumerate(cl):
                      pending[c] = (u[k], pn[k])
                  return sel, u, pn, None

              @staticmethod
              def done(idx, res, accepted):
                  ok = np.ones(len(idx), dtype=bool)
                  books(idx, ok, res["Hcur"], res["Hnew"], res["Unew"], res["x"], res.get("dsyn_new"), None, acc=accepted)
                  for c in idx:
                      pending.pop(int(c), None)
                  return dict(dt=dt[idx], rem=sampler._traj_len(dt[idx]))

              @staticmethod
              def withdraw(idx):                      # nothing to rewind: process_done uses the early draws
                  pass

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
              break                                     # stopped inside a segment: no barrier state to keep
          if target is None:
              break
          cur = target
          if every and cur % every == 0 and np.any(i < total):
              self._save_checkpoint(x, i, ncount, misfit, x_cache, syndata, dt, dtbar, h0, cur)
        self.finished = not bool(np.any(i < total))
        self.naccepted, self.ntrajectories = i.copy(), ncount.copy()
        if not self.finished:                    # stopped by max_steps: nothing is written
            self.x_cache, self.dt_final = x_cache, dt
            return misfit[0] if nc == 1 else misfit
        return self._finish(misfit, x_cache, syndata, i, ncount, dt)

    def _save_checkpoint(self, x, i, ncount, misfit, x_cache, syndata, dt, dtbar, h0, ntraj):
        if hasattr(self.model, "reset_warm_start"):      # the run may be cut here: what follows starts from the full search, as a resumed run would
            self.model.reset_warm_start()
        save_checkpoint(self.checkpoint, self.rng, x=x, i=i, ncount=ncount, misfit=misfit, x_cache=x_cache,
                        syndata=syndata, initmodel=self.initmodel, ii=self.ii, dt=dt, dtbar=dtbar, h0=h0, ntraj=ntraj,
                        inverse_mass=self.inverse_mass)
