"""Shared host logic of the batched samplers: per-chain legacy numpy RNG streams that reproduce
the reference's draw order (pyhmc/hmc.py:43,61: ``np.random.seed(seed + rank)``; chain c here ==
MPI rank c there), initial models, result store (per-chain files with the reference's member names, one
batched file per rank, exporter between the two) and checkpoint / resume of a running sampler."""
import contextlib
import os

import numpy as np


def _load_rngbatch():
    """librngbatch.so (csrc_host/rngbatch.c, built by rfsurfhmc_amd.build): C loop over the chains' streams.  It
    advances the MT19937 states inside numpy in place, so it is only used when numpy's bit generator is the expected
    one (struct of 624 key words + position): _rngbatch_selftest compares it with rs.rand() / rs.randn() / get_state()
    on scratch streams before the first use and the Python path takes over on any mismatch."""
    import ctypes
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "librngbatch.so")
    if not os.path.exists(path):
        return None
    try:
        L = ctypes.CDLL(path)
    except OSError:
        return None
    vp, i64, u64 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64
    L.rngbatch_randn.argtypes = [vp, vp, vp, vp, i64, i64, vp]
    L.rngbatch_randn.restype = None
    L.rngbatch_rand.argtypes = [vp, vp, i64, vp]
    L.rngbatch_rand.restype = None
    for f in (L.rngbatch_save, L.rngbatch_load):
        f.argtypes, f.restype = [vp, vp, i64, vp], None
    return L if _rngbatch_selftest(L) else None


_RNGBATCH_OK = {}


def _rngbatch_selftest(L):
    """librngbatch.so rewrites numpy's private MT19937 state in place under an assumed layout (624 key words + position
    behind BitGenerator.ctypes.state_address).  Before it is trusted, scratch streams are driven through it and compared
    with numpy's own draws, odd counts (the cached second Gaussian), save / load and get_state included; any mismatch --
    another numpy, a stale library -- and the Python path is used instead (same numbers, slower)."""
    key = id(L)
    if key in _RNGBATCH_OK:
        return _RNGBATCH_OK[key]
    ok = False
    try:
        seeds = (12345, 991206)
        a = [np.random.RandomState(sd) for sd in seeds]             # driven by the library
        b = [np.random.RandomState(sd) for sd in seeds]             # driven by numpy
        bgs = [r._bit_generator for r in a]
        states = np.array([g.ctypes.state_address for g in bgs], dtype=np.uint64)
        has = np.zeros(2, dtype=np.int32); gauss = np.zeros(2)
        idx = np.arange(2, dtype=np.int64)
        P = lambda x: x.ctypes.data
        ok = True
        for n in (3, 1, 4):                                         # odd counts leave a cached deviate behind
            out = np.empty((2, n)); L.rngbatch_randn(P(states), P(has), P(gauss), P(idx), 2, n, P(out))
            ok = ok and np.array_equal(out, np.stack([r.randn(n) for r in b]))
            u = np.empty(2); L.rngbatch_rand(P(states), P(idx), 2, P(u))
            ok = ok and np.array_equal(u, np.array([r.rand() for r in b]))
        buf = np.empty((2, 625), dtype=np.uint32); L.rngbatch_save(P(states), P(idx), 2, P(buf))
        for k in range(2):                                          # the saved words are numpy's own state
            st = b[k].get_state()
            ok = ok and np.array_equal(buf[k, :624], st[1]) and int(buf[k, 624]) == int(st[2])
            ok = ok and int(has[k]) == int(st[3]) and (not st[3] or float(gauss[k]) == float(st[4]))
        u = np.empty(2); L.rngbatch_rand(P(states), P(idx), 2, P(u))        # advance, rewind, draw again
        L.rngbatch_load(P(states), P(idx), 2, P(buf))
        u2 = np.empty(2); L.rngbatch_rand(P(states), P(idx), 2, P(u2))
        ok = ok and np.array_equal(u, u2) and np.array_equal(u, np.array([r.rand() for r in b]))
        ok = ok and all(np.array_equal(a[k].get_state()[1], b[k].get_state()[1]) for k in range(2))
    except Exception:
        ok = False
    _RNGBATCH_OK[key] = bool(ok)
    return _RNGBATCH_OK[key]


class ChainRNG:
    """One legacy MT19937 stream per chain: RandomState(seed + first_chain + c) draws exactly what the
    reference's global generator draws on rank (first_chain + c).

    With librngbatch.so the uniform and normal deviates of many chains are drawn by one C call that drives each
    stream's own bit generator (same values as rs.rand() / rs.randn(n), bit for bit); the cached second deviate of
    numpy's legacy Gaussian then lives in this object (has_gauss / gauss) instead of inside the RandomState.  Without
    the library the same draws are made chain by chain in Python."""

    def __init__(self, seed, first_chain, nchains):
        import ctypes
        self.rs = [np.random.RandomState(seed + first_chain + c) for c in range(nchains)]
        self._L = _load_rngbatch()
        if self._L is not None:
            bgs = [r._bit_generator for r in self.rs]
            self._bgs = bgs                                   # keeps the ctypes interfaces (and their states) alive
            self._states = np.array([b.ctypes.state_address for b in bgs], dtype=np.uint64)
            self._has = np.zeros(nchains, dtype=np.int32)
            self._gauss = np.zeros(nchains)

    @staticmethod
    def _p(a):
        return a.ctypes.data

    def rand(self, idx):
        if self._L is None:
            return np.array([self.rs[c].rand() for c in idx])
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        out = np.empty(len(idx))
        if len(idx):
            self._L.rngbatch_rand(self._p(self._states), self._p(idx), len(idx), self._p(out))
        return out

    def randn(self, idx, n):
        if self._L is None:
            return np.stack([self.rs[c].randn(n) for c in idx]) if len(idx) else np.zeros((0, n))
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        out = np.empty((len(idx), n))
        if len(idx):
            self._L.rngbatch_randn(self._p(self._states), self._p(self._has), self._p(self._gauss), self._p(idx),
                                   len(idx), n, self._p(out))
        return out

    def snapshot(self, idx):
        """Everything needed to put the streams of chains ``idx`` back where they are now (see restore)."""
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        if self._L is None:
            return idx, [self.rs[c].get_state() for c in idx]
        buf = np.empty((len(idx), 625), dtype=np.uint32)
        if len(idx):
            self._L.rngbatch_save(self._p(self._states), self._p(idx), len(idx), self._p(buf))
        return idx, (buf, self._has[idx].copy(), self._gauss[idx].copy())

    def restore(self, snap, which):
        """Rewind the streams of the chains ``which`` (a subset of the snapshot's chains) to the snapshot."""
        idx, data = snap
        pos = {int(c): k for k, c in enumerate(idx)}
        rows = np.array([pos[int(c)] for c in which], dtype=np.int64)
        sel = np.ascontiguousarray(idx[rows])
        if self._L is None:
            for c, k in zip(sel, rows):
                self.rs[int(c)].set_state(data[k])
            return
        buf, has, gauss = data
        b = np.ascontiguousarray(buf[rows])
        if len(sel):
            self._L.rngbatch_load(self._p(self._states), self._p(sel), len(sel), self._p(b))
            self._has[sel] = has[rows]; self._gauss[sel] = gauss[rows]

    def randint(self, idx, lo, hi):
        return np.array([self.rs[c].randint(lo, hi) for c in idx], dtype=np.int32)

    def get_state(self):
        """Arrays that restore every stream exactly (MT19937 key, position, cached Gaussian)."""
        st = [r.get_state() for r in self.rs]
        d = {"rng_key": np.stack([t[1] for t in st]), "rng_pos": np.array([t[2] for t in st])}
        if self._L is None:
            d["rng_has_gauss"] = np.array([t[3] for t in st]); d["rng_gauss"] = np.array([t[4] for t in st])
        else:
            d["rng_has_gauss"] = self._has.copy(); d["rng_gauss"] = self._gauss.copy()
        return d

    def set_state(self, d):
        for c, r in enumerate(self.rs):
            keep = self._L is None
            r.set_state(("MT19937", d["rng_key"][c], int(d["rng_pos"][c]), int(d["rng_has_gauss"][c]) if keep else 0,
                         float(d["rng_gauss"][c]) if keep else 0.0))
        if self._L is not None:
            self._has[:] = np.asarray(d["rng_has_gauss"], dtype=np.int32)
            self._gauss[:] = np.asarray(d["rng_gauss"], dtype=np.float64)


def set_initial_model(rs, boundaries):
    """pyhmc/hmc.py:74-93 == hmcda.py:99-124: uniform in bounds, vs sorted ascending, thk permuted alike."""
    n = boundaries.shape[0]
    xcur = np.zeros(n)
    for i in range(n):
        bdl, bdr = boundaries[i, 0], boundaries[i, 1]
        xcur[i] = bdl + (bdr - bdl) * rs.rand()
    half = n // 2
    idx = np.argsort(xcur[:half])
    xcur[:half] = xcur[:half][idx]
    xcur[half:] = xcur[half:][idx]
    return xcur


def check_init_is_in_boundary(xcur, boundaries):
    """pyhmc/hmc.py:95-99 (the last entry is not checked)."""
    for i in range(len(xcur) - 1):
        if xcur[i] < boundaries[i, 0] or xcur[i] > boundaries[i, 1]:
            return False
    return True


def initial_models(rng: ChainRNG, boundaries):
    xs = []
    for rs in rng.rs:
        x = set_initial_model(rs, boundaries)
        while not check_init_is_in_boundary(x, boundaries):
            x = set_initial_model(rs, boundaries)
        xs.append(x)
    return np.stack(xs)


def store_format(fmt="auto"):
    """"h5" or "npz".  ``auto`` is the reference's HDF5 wherever it can be written (h5py, or libhdf5 through
    pyhmc/_h5.py) and .npz otherwise; asking for "h5" where it cannot be written raises ImportError."""
    from . import _h5
    if fmt in (None, "auto"):
        return "h5" if _h5.backend() else "npz"
    if fmt not in ("h5", "npz"):
        raise ValueError("store format should be auto, h5 or npz")
    if fmt == "h5" and _h5.backend() is None:
        _h5.library()                       # raises the ImportError that says where it looked
    return fmt


def save_chain_results(outdir, name, rank, initmodel, obs, xmean, synmean, x_cache, syndata, fmt="npz"):
    """One chain's results under the reference's member names (pyhmc/hmc.py:203-226, 272-275).
    ``fmt="h5"``: {name}.{rank}.h5 exactly as the reference lays it out -- datasets initmodel, obs, groups
    mean/{model,syn} and {i}/{model,syn} for every sample i -- the file src/plot_results.py:106-156 reads.
    ``fmt="npz"``: {name}.{rank}.npz with keys initmodel, obs, mean/model, mean/syn, model[i], syn[i]."""
    os.makedirs(outdir, exist_ok=True)
    if fmt == "h5":
        from . import _h5
        path = os.path.join(outdir, f"{name}.{rank}.h5")
        with _h5.open_file(path, "w") as f:
            f.create_dataset("initmodel", data=np.asarray(initmodel, dtype=np.float64))
            f.create_dataset("obs", data=np.asarray(obs, dtype=np.float64))
            groups = [("mean", xmean, synmean)] + [(str(i), x_cache[i], None if syndata is None else syndata[i])
                                                   for i in range(len(x_cache))]
            for g, x, syn in groups:
                f.create_group(g)
                f.create_dataset(f"{g}/model", data=np.asarray(x, dtype=np.float64))
                if syn is not None:
                    f.create_dataset(f"{g}/syn", data=np.asarray(syn, dtype=np.float64))
        return path
    d = {"initmodel": initmodel, "obs": obs, "mean/model": xmean, "mean/syn": synmean, "model": x_cache}
    if syndata is not None:
        d["syn"] = syndata
    path = os.path.join(outdir, f"{name}.{rank}.npz")
    np.savez(path, **d)
    return path


def load_chain_results(path):
    """Either per-chain format back as one dict: initmodel, obs, mean/model, mean/syn, model [ns, nx], syn [ns, nd]
    (syn missing when it was not stored)."""
    if path.endswith(".npz"):
        z = np.load(path)
        return {k: z[k] for k in z.files}
    from . import _h5
    with _h5.open_file(path, "r") as f:
        d = {k: f[k][:] for k in ("initmodel", "obs", "mean/model", "mean/syn")}
        ns = 0
        while str(ns) in f:
            ns += 1
        d["model"] = np.stack([f[f"{i}/model"][:] for i in range(ns)]) if ns else np.zeros((0, len(d["initmodel"])))
        if ns and f"0/syn" in f:
            d["syn"] = np.stack([f[f"{i}/syn"][:] for i in range(ns)])
    return d


_BATCHED_KEYS = ("first_chain", "initmodel", "obs", "mean_model", "mean_syn", "model", "syn", "misfit")


def save_batched_results(outdir, name, rank, first_chain, initmodel, obs, xmean, synmean, x_cache, syndata, misfit,
                         fmt="npz"):
    """One file per rank, every chain inside ({name}.rank{rank}.npz or .h5): first_chain, initmodel [nc, nx],
    obs [nd], mean_model [nc, nx], mean_syn [nc, nd], model [nc, ns, nx], syn [nc, ns, nd] (optional),
    misfit [nc, ns].  The reference writes one HDF5 per MPI rank = per chain (pyhmc/hmc.py:203-226, 272-275);
    with thousands of chains per GPU that is the wrong granularity -- export_chain() recreates a single chain's
    file on demand.  In the HDF5 form the datasets are contiguous and chain-major, so one chain's samples are one
    hyperslab (``f["model"][c]``)."""
    os.makedirs(outdir, exist_ok=True)
    d = {"first_chain": np.array(first_chain, dtype=np.int64), "initmodel": initmodel, "obs": obs, "mean_model": xmean,
         "mean_syn": synmean, "model": x_cache, "misfit": misfit}
    if syndata is not None:
        d["syn"] = syndata
    if fmt == "h5":
        from . import _h5
        path = os.path.join(outdir, f"{name}.rank{rank}.h5")
        with _h5.open_file(path, "w") as f:
            for k, v in d.items():
                f.create_dataset(k, data=np.asarray(v))
        return path
    path = os.path.join(outdir, f"{name}.rank{rank}.npz")
    np.savez(path, **d)
    return path


class _BatchedView:
    """Read access to a batched result file of either format: ``v[key]`` whole arrays, ``v.chain(key, c)`` one
    chain's block (a single hyperslab read in the HDF5 form)."""

    def __init__(self, path):
        self.path = path
        if path.endswith(".npz"):
            self._z, self._f = np.load(path), None
            self.files = list(self._z.files)
        else:
            from . import _h5
            self._z, self._f = None, _h5.open_file(path, "r")
            self.files = [k for k in _BATCHED_KEYS if k in self._f]

    def __getitem__(self, k):
        return self._z[k] if self._f is None else self._f[k][()] if self._f[k].shape == () else self._f[k][:]

    def chain(self, k, c):
        return self._z[k][c] if self._f is None else self._f[k][c]

    def nchain(self):
        return (self._z["model"] if self._f is None else self._f["model"]).shape[0]

    def close(self):
        if self._f is not None:
            self._f.close()


def load_batched_results(path):
    """A batched result file (.npz or .h5) as a dict of arrays."""
    v = _BatchedView(path)
    try:
        return {k: np.asarray(v[k]) for k in v.files}
    finally:
        v.close()


def export_chain(batched_path, chain, outdir=None, name=None, fmt="auto"):
    """Write chain ``chain`` (global chain number = the reference's MPI rank) of a batched result file (.npz or
    .h5) in the reference's per-rank layout, see save_chain_results: ``fmt="h5"`` -> {name}.{chain}.h5,
    ``fmt="npz"`` -> {name}.{chain}.npz, ``"auto"`` -> HDF5 where it can be written."""
    fmt = store_format(fmt)
    z = _BatchedView(batched_path)
    try:
        c = int(chain) - int(z["first_chain"])
        if not 0 <= c < z.nchain():
            raise IndexError(f"chain {chain} is not in {batched_path}")
        outdir = os.path.dirname(batched_path) if outdir is None else outdir
        name = os.path.basename(batched_path).split(".rank")[0] if name is None else name
        syn = z.chain("syn", c) if "syn" in z.files else None
        return save_chain_results(outdir, name, chain, z.chain("initmodel", c), z["obs"], z.chain("mean_model", c),
                                  z.chain("mean_syn", c), z.chain("model", c), syn, fmt=fmt)
    finally:
        z.close()


def ensemble_inverse_mass(x, clip=(1e-3, 1e3)):
    """Diagonal M^-1 for the samplers from the spread of the chains themselves: the cross-chain variance of the
    current models (thousands of chains make a history unnecessary), pooled over ranks, normalised to geometric
    mean 1 so that the step size keeps its meaning, and clipped.  Parameters on which the chains do not differ
    (variance 0) get the lower clip."""
    from ..chains import pooled_variance
    var = pooled_variance(x)
    pos = var > 0
    if not pos.any():
        return np.ones_like(var)
    g = np.exp(np.mean(np.log(var[pos])))
    return np.clip(np.where(pos, var / g, clip[0]), clip[0], clip[1])


def save_checkpoint(path, rng: ChainRNG, **state):
    """Everything a sampler needs to continue bit-for-bit: its arrays + every chain's RNG stream.
    Written to a temporary file first, then renamed (a killed job never leaves a torn checkpoint)."""
    d = {k: np.asarray(v) for k, v in state.items() if v is not None}
    d.update(rng.get_state())
    tmp = path + ".tmp.npz"
    np.savez(tmp, **d)
    os.replace(tmp, path)


def load_checkpoint(path, rng: ChainRNG):
    z = np.load(path)
    rng.set_state(z)
    return {k: z[k] for k in z.files if not k.startswith("rng_")}


def cpu_quota():
    """CPUs this process may really use: its affinity mask capped by the cgroup CPU quota (cpu.max, or the v1 pair)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", ):
        try:
            q, per = open(path).read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(int(q) / int(per))))
        except (OSError, ValueError):
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, q // per))
    except (OSError, ValueError):
        pass
    return n


@contextlib.contextmanager
def host_threads(limit=None):
    """Cap torch's intra-op CPU threads while the host side of a sampler runs, and restore them afterwards.
    The host work of a flow step is a handful of small copies; with torch's default (one OpenMP thread per visible CPU)
    every such copy wakes the whole pool, whose idle threads then spin -- under a cgroup CPU quota (16 of 256 CPUs on the
    GPU boxes used here) that burns the quota within half of each 100 ms period and the kernel freezes the process for
    the rest of it: measured 52 ms stalls every ~3 device steps, 37.7 -> 17.9 ms per DA flow step once capped."""
    import torch
    before = torch.get_num_threads()
    cap = max(1, min(before, int(limit) if limit else min(4, cpu_quota())))
    torch.set_num_threads(cap)
    try:
        yield cap
    finally:
        torch.set_num_threads(before)


def with_host_threads(fn):
    """Decorator form of host_threads() for the samplers' entry points."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        with host_threads():
            return fn(*a, **k)
    return wrapped


_SIDE_STREAMS = {}
# How the HOST waits for an event.  A rank runs one step ahead of the device and waits for most of every step; the runtime's
# synchronize calls spin -- measured on the GPU boxes (ROCm 7.2): Event.synchronize() keeps the calling thread at 100 % of a
# core, with or without the blocking-sync flag, and a helper thread of the runtime beside it: 2.0 cores per rank, 9.1 ms of CPU
# per 4.65 ms step (bench.py: host_cpu_ms_per_step), which eight ranks on a node's 16-CPU quota do not have.  Polling
# Event.query() between short sleeps costs 0.002 s of CPU where the spin costs 0.143 s.  RFS_HOST_SPIN=1: the runtime's wait.
HOST_WAIT_BLOCKS = False
HOST_WAIT_POLLS = os.environ.get("RFS_HOST_SPIN", "0") != "1"
HOST_POLL_SLEEP = 5.0e-5


def host_wait(ev):
    """Wait for a torch.cuda.Event on the host without spinning (see HOST_WAIT_POLLS)."""
    if HOST_WAIT_POLLS:
        import time
        while not ev.query():
            time.sleep(HOST_POLL_SLEEP)
    else:
        ev.synchronize()
ASYNC_MIN_CHAINS = int(os.environ.get("RFS_ASYNC_MIN_CHAINS", "64"))      # populations below this keep hand-backs in the foreground (run_flow)


LEGACY_TAIL = os.environ.get("RFS_FLOW_LEGACY_TAIL") == "1"      # (A/B runs: round 4's staging copy + immediate wait between two steps)


def _side_stream(dev):
    import torch
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[key]


def run_flow(model, st, process_done, active, fetch_syn=True, pipeline=True, max_steps=None, step_hook=None,
             restart=None, async_handback=True):
    """Drive model.flow_step (rfs_flow_step) until ``active()`` is False.

    After every step the chains that finished a trajectory are handed to ``process_done(idx, res)`` (host side:
    accept / reject, bookkeeping, RNG draws), which returns ``(xkeep[len(idx), nx], restart)`` with ``restart`` either
    None or a dict(idx=..., p=..., rem=..., dt=optional) for the chains that start another trajectory.
    pipeline=True: step s+1 is launched BEFORE the results of step s are processed, so the host work (one Python
    loop over the finished chains with per-chain RNG streams) overlaps the GPU step; the chains that finished at
    step s sit out step s+1 and restart with step s+2.  Each chain still sees exactly the same sequence of
    trajectories and draws, so the samples do not depend on ``pipeline``.  Returns the number of device steps.
    max_steps: stop after that many device steps (the run is then unfinished: benchmarks, smoke runs);
    step_hook(s, st): called right before device step s is launched (s = 0, 1, ...; bench.py takes its time stamps
    and counts the chains inside a trajectory there).
    restart: restarts on the device (rfs_flow_step2), for samplers whose draws do not depend on the trajectory.  An object
    with ``rem0`` (lengths of the trajectories the state starts with), ``predraw(chains) -> (sel, u, p, rem)`` (for a
    subset ``sel`` of ``chains`` -- the ones that complete their trajectory in the step launched next -- the acceptance
    draw and the next trajectory's momentum and length, drawn now), ``done(idx, res, accepted)`` (books of chains the
    device accepted / rejected and restarted; ``res`` holds Ucur, Hcur, Hnew, Unew, x [, dsyn_new] of the completed
    trajectory) and ``withdraw(idx)`` (chains that failed with a deposit outstanding: rewind their streams, they come
    through ``process_done`` next).  Such chains evaluate their new start model in the very next step instead of sitting
    one out; the sequence of draws and decisions per chain is unchanged.
    async_handback: rfs_set_option "flow_async_handback" for the run -- a chain whose root search is handed back to the
    reference-semantics search sits that device step out (the search runs beside the next step) instead of making every
    chain wait for it; the chain's own sequence of models and decisions is unchanged."""
    import torch
    dev = st["x"].device
    # (events and side streams below belong to the state's device, whichever device is current in the caller)
    ctx = getattr(model, "_ctx", None)
    # (a handful of chains: a chain that sits steps out while its search runs in the background leaves the device with
    # nothing to do -- every such step still costs its dozen launches, 0.75 ms for ONE chain -- so small populations take
    # the search in the foreground: configs[0], one chain at dt 0.1, 3.2 -> 1.8 ms per evaluation)
    if st["x"].shape[0] < ASYNC_MIN_CHAINS:
        async_handback = False
    if ctx is not None and dev.type == "cuda":
        ctx.set_option("flow_async_handback", int(bool(async_handback) and os.environ.get("RFS_FLOW_ASYNC", "1") != "0"))
    try:
        with host_threads(), (torch.cuda.device(dev) if dev.type == "cuda" else contextlib.nullcontext()):
            return _run_flow(model, st, process_done, active, fetch_syn, pipeline, max_steps, step_hook, restart)
    finally:
        if ctx is not None and dev.type == "cuda":
            ctx.set_option("flow_async_handback", 0)          # (direct callers of flow_step count on one step per call)


def _run_flow(model, st, process_done, active, fetch_syn, pipeline, max_steps, step_hook, restart):
    import torch
    dev = st["x"].device

    nchain_all = st["x"].shape[0]
    pinned = {}                                  # (trailing shape, dtype) -> [rotating pinned buffers, next]

    def t(a):
        """Host array (at most one row per chain) -> device tensor.  On a GPU the copy goes through preallocated pinned
        staging buffers and is asynchronous: a plain .to(device) of pageable memory is stream-ordered behind the step that
        was just launched AND blocks the host until it has run, which would serialise the host bookkeeping with the GPU
        step it is meant to overlap.  Eight buffers per shape rotate; each carries an event recorded on the stream that
        issued its last copy (the main stream or the side stream of the early deposits), and a buffer is only refilled once
        that event has completed -- however many uploads an iteration makes."""
        a = np.ascontiguousarray(a)
        if dev.type != "cuda":
            return torch.from_numpy(a).to(dev)
        src = torch.from_numpy(a)
        key = (tuple(a.shape[1:]), src.dtype)
        if key not in pinned:
            pinned[key] = [[[torch.empty((nchain_all,) + key[0], dtype=src.dtype, pin_memory=True), None] for _ in range(8)], 0]
        bufs, nxt = pinned[key]
        pinned[key][1] = (nxt + 1) % len(bufs)
        slot = bufs[nxt]
        if slot[1] is not None:
            host_wait(slot[1])                    # the copy that last read this buffer has finished
        h = slot[0][: a.shape[0]]
        h.copy_(src)
        out = h.to(dev, non_blocking=True)
        if slot[1] is None:
            slot[1] = torch.cuda.Event(blocking=HOST_WAIT_BLOCKS)
        slot[1].record(torch.cuda.current_stream(dev))
        return out

    # On a GPU with pipeline=True the results of step s are fetched on a side stream WHILE step s+1 runs: step s+1 is
    # launched first, the side stream waits for the event recorded behind step s and gathers the finished chains' rows
    # (those chains idle in step s+1, so nothing writes them; the done flags alternate between two buffers because
    # every step clears its own).  The device then never waits for the host between steps.
    # (ONE side stream per device and process: torch deals its pool streams round-robin, and a stream that lands on the
    # hardware queue of the library's background-search stream has its small copies queue behind 4 ms searches -- a second
    # or third run in a process was then twice as slow as the first: measured, 5.8 vs 11.5 ms per step)
    side = _side_stream(dev) if (dev.type == "cuda" and pipeline) else None
    dbuf = [st["done"], torch.zeros_like(st["done"])]
    marks = []                                   # (done buffer, event, step index) of the steps not fetched yet
    if restart is not None and not hasattr(model, "flow_restart_state"):
        restart = None
    if restart is not None:
        deferred = bool(getattr(restart, "deferred", False))          # lengths / step sizes follow a call later (done())
        model.flow_restart_state(st, want_dsyn=fetch_syn, deferred=deferred)
        finish = np.asarray(restart.rem0, dtype=np.int64).copy()      # step in which each chain completes (fresh at step 0)
        has_dep = np.zeros(nchain_all, dtype=bool)                    # deposit outstanding
        dep_rem = np.zeros(nchain_all, dtype=np.int64)                # length of the deposited trajectory

    # Records (rfs_flow_step3, round 6): the device packs what the books need of every chain that completed a trajectory into a
    # pinned buffer the host reads behind the step's event -- no copy of the flags down, no index lists up, no gathers: the
    # per-step fetch costs the device nothing (before: 7-15 copies and 2-8 index kernels per step on the side stream)
    nx_ = st["x"].shape[1]
    use_rec = (dev.type == "cuda" and hasattr(model, "flow_deposit") and os.environ.get("RFS_FLOW_RECORDS", "1") != "0"
               and hasattr(getattr(getattr(model, "_ctx", None), "L", None), "rfs_flow_step3"))
    if use_rec:
        rstride = 8 + nx_ + (st["dsyn_new"].shape[1] if fetch_syn else 0)
        rcap = 3 * nchain_all                     # a ring: the step being read, the one under way, and slack
        ring = torch.zeros(rcap * rstride, dtype=torch.float64, pin_memory=True)
        ring_h = ring.numpy().reshape(rcap, rstride)
        rd = [0]                                  # read cursor: records consumed since the ring started

    def fetch():
        """-> (s, idx1, res1, idx2, res2, acc2): step index; chains that finished and wait for the host (done = 1) with
        their rows; chains the device restarted (done = 2 / 3) with the parked results and the accept flags."""
        done, ev, s, rb = marks.pop(0)
        if rb is not None:
            host_wait(ev)                         # the step is through: its records are complete
            stamp = float(s + 1)
            # this step's records: from the cursor on, as long as they carry its stamp (then comes a stale slot or one of the
            # step that is running now)
            k0 = rd[0] % rcap
            st7 = np.concatenate((ring_h[k0:, 7], ring_h[:k0, 7]))[:nchain_all + 1]
            nrec = int(np.argmax(st7 != stamp)) if (st7 != stamp).any() else len(st7)
            assert nrec <= nchain_all, nrec
            sl = (k0 + np.arange(nrec)) % rcap
            rd[0] += nrec
            R = ring_h[sl]                                          # (a copy: the slots are rewritten a lap later)
            R = R[np.argsort(R[:, 0], kind="stable")]
            chain, code = R[:, 0].astype(np.int64), R[:, 1].astype(np.int64)
            out = []
            for m in (code == 1, code >= 2):
                r = R[m]
                res = dict(ok=r[:, 2].astype(np.int32), Ucur=r[:, 3].copy(), Hcur=r[:, 4].copy(), Hnew=r[:, 5].copy(),
                           Unew=r[:, 6].copy(), x=r[:, 8:8 + nx_].copy()) if len(r) else None
                if res is not None and fetch_syn:
                    res["dsyn_new"] = r[:, 8 + nx_:].copy()
                out.append((chain[m], res))
            return s, out[0][0], out[0][1], out[1][0], out[1][1], code[code >= 2] == 3
        if side is not None:
            side.wait_event(ev)
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            d = done.cpu().numpy()
            idx1, idx2 = np.nonzero(d == 1)[0], np.nonzero(d >= 2)[0]
            res1 = res2 = None
            if len(idx1):
                sel = t(idx1)
                keys = ["ok", "Hcur", "Hnew", "Unew", "Ucur", "x"] + (["dsyn_new"] if fetch_syn else [])
                res1 = {k: st[k].index_select(0, sel).cpu().numpy() for k in keys}
            if len(idx2):
                sel = t(idx2)
                val = st["res_val"].index_select(0, sel).cpu().numpy()
                res2 = dict(Ucur=val[:, 0], Hcur=val[:, 1], Hnew=val[:, 2], Unew=val[:, 3],
                            x=st["res_x"].index_select(0, sel).cpu().numpy())
                if fetch_syn:
                    res2["dsyn_new"] = st["res_dsyn"].index_select(0, sel).cpu().numpy()
        return s, idx1, res1, idx2, res2, d[idx2] == 3

    uploaded = []                                # event behind the previous iteration's copies out of the pinned buffers

    def apply(idx, xkeep, rs_):
        st["x"].index_copy_(0, t(idx), t(xkeep))
        if rs_ is not None and len(rs_["idx"]):
            rs = t(np.asarray(rs_["idx"]))
            st["p"].index_copy_(0, rs, t(rs_["p"]))
            st["rem"].index_copy_(0, rs, t(np.asarray(rs_["rem"], dtype=np.int32)))
            if rs_.get("dt") is not None:
                st["dt"].index_copy_(0, rs, t(np.asarray(rs_["dt"], dtype=np.float64)))
            st["fresh"].index_fill_(0, rs, 1)
            st["ok"].index_fill_(0, rs, 1)

    # The same in ONE copy and ONE launch (rfs_flow_restart) where the model offers it: a dozen small copies and scatters on
    # the main stream between two steps cost the device 0.3 ms of every step in which some chain goes through the host
    # (a failed evaluation ends a trajectory without an acceptance draw: a few chains in most steps of a real run)
    fused = dev.type == "cuda" and hasattr(model, "flow_restart")
    stage = [[None, None] for _ in range(8)]      # rotating pinned byte buffers: [tensor, event behind its last copy]
    stage_next = [0]

    ZERO_COPY_MAX = 1 << 16                       # bytes a launch reads straight from pinned memory; above: one async copy first

    def staged(parts_of):
        """Pack host arrays into ONE pinned byte buffer.  parts_of(put) calls put(array, dtype) -> byte offset for each.
        Returns (slot, source tensor for the launch): the pinned buffer itself when the lists are small (pinned host memory is
        mapped into the device's address space at the same address: a few KB over the link cost a kernel less than a copy
        kernel in front of it did, profiles/r04_step_timeline.txt), a device copy of it (ONE async copy on the current
        stream) above ZERO_COPY_MAX -- restart=None runs put every finished chain's row through here, MBs at 8192 chains."""
        parts, off = [], [0]

        def put(a, dtype):
            a = np.ascontiguousarray(a, dtype=dtype)
            o = off[0]
            parts.append((o, a))
            off[0] = (o + a.nbytes + 7) & ~7
            return o

        offs = parts_of(put)
        slot = stage[stage_next[0]]; stage_next[0] = (stage_next[0] + 1) % len(stage)
        if slot[1] is not None:
            host_wait(slot[1])                    # the launch / copy that last read this buffer has run
        if slot[0] is None or slot[0].numel() < off[0]:
            slot[0] = torch.empty(max(off[0], 1 << 16), dtype=torch.uint8, pin_memory=True)
        h = slot[0].numpy()
        for o, a in parts:
            h[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)
        if slot[1] is None:
            slot[1] = torch.cuda.Event(blocking=HOST_WAIT_BLOCKS)
        src = slot[0] if (off[0] <= ZERO_COPY_MAX and not LEGACY_TAIL) else slot[0][:off[0]].to(dev, non_blocking=True)
        return slot, src, offs

    def apply_fused(idx, xkeep, rs_, wd):
        def parts_of(put):
            n1 = len(idx)
            o_idx1 = o_xk = None
            if n1:
                o_idx1 = put(idx, np.int32); o_xk = put(xkeep, np.float64)
            n2 = 0; o_idx2 = o_p = o_rem = o_dt = None
            if rs_ is not None and len(rs_["idx"]):
                n2 = len(rs_["idx"])
                o_idx2 = put(rs_["idx"], np.int32); o_rem = put(rs_["rem"], np.int32)
                if rs_.get("p") is not None:         # (absent: lengths and step sizes of chains already under way)
                    o_p = put(rs_["p"], np.float64)
                if rs_.get("dt") is not None:
                    o_dt = put(rs_["dt"], np.float64)
            n3 = len(wd)
            o_idx3 = put(wd, np.int32) if n3 else None
            return n1, o_idx1, o_xk, n2, o_idx2, o_p, o_rem, o_dt, n3, o_idx3
        slot, src, offs = staged(parts_of)
        model.flow_restart(st, src, *offs)
        slot[1].record(torch.cuda.current_stream(dev))            # the buffer is free again once the launch has run

    steps = 0

    deposited = []                               # events behind deposits made on the side stream
    withdrawn = []                               # events behind rfs_flow_restart launches that withdrew deposits

    def step():
        nonlocal steps
        while deposited:
            torch.cuda.current_stream().wait_event(deposited.pop())
        if step_hook is not None:
            step_hook(steps, st)
        st["done"] = dbuf[steps % 2]
        rb = ring if use_rec else None
        st["rec"] = (ring, rcap, fetch_syn, float(steps + 1), steps == 0) if use_rec else None
        model.flow_step(st); steps += 1
        ev = None
        if side is not None or use_rec:
            ev = torch.cuda.Event(blocking=HOST_WAIT_BLOCKS); ev.record()
        marks.append((st["done"], ev, steps - 1, rb))

    capped = lambda: max_steps is not None and steps >= max_steps
    step()
    while True:
        early = side is not None and active() and not capped()
        if early:
            step()                               # before the fetch: see above
        s, idx1, res1, idx2, res2, acc2 = fetch()            # synchronises with the step whose results it takes
        more = (active() or len(idx1) + len(idx2) > 0) and not capped()
        if pipeline and more and not early:
            step()                               # the chains that wait for the host idle in this step (rem = -1, fresh = 0)
        if uploaded:
            host_wait(uploaded.pop())            # bounds how far the host runs ahead (the staging buffers guard themselves, t())
        if len(idx1):
            wd = idx1[:0]
            if restart is not None:
                wd = idx1[has_dep[idx1]]
                if len(wd):                      # failed with a deposit outstanding: take it back before anything is drawn
                    restart.withdraw(wd)
                    has_dep[wd] = False
                    if not fused:
                        st["nxt_have"].index_fill_(0, t(wd), 0)
                        if side is not None:     # a deposit made later on the side stream must land behind this fill
                            wev = torch.cuda.Event(); wev.record(); side.wait_event(wev)
            xkeep, rs_ = process_done(idx1, res1)
            if fused:
                apply_fused(idx1, xkeep, rs_, wd)            # stream-ordered after the step launched above
                if len(wd) and side is not None:
                    # A deposit made LATER for a withdrawn chain must land behind this launch (it clears the chain's `have`).
                    # Such a chain restarts with a trajectory of its own first: nothing is deposited for it before the
                    # iteration after next -- so the side stream takes the dependency at the NEXT iteration's deposits, when
                    # the launch has long run.  (Waiting here made this iteration's deposits, and with them the next step's
                    # first kernel, wait for a launch that sits behind the whole step under way: 0.17 ms of idle device
                    # between two steps, profiles/r04_step_timeline.txt.)
                    wev = torch.cuda.Event(); wev.record(); withdrawn.append((steps, wev, np.asarray(wd)))
                    if LEGACY_TAIL:
                        side.wait_event(withdrawn.pop()[1])
            else:
                apply(idx1, xkeep, rs_)
            if restart is not None:
                finish[idx1] = -1
                if rs_ is not None and len(rs_["idx"]):              # fresh in the step launched next
                    finish[np.asarray(rs_["idx"])] = steps + np.asarray(rs_["rem"], dtype=np.int64)
        if len(idx2):
            late = restart.done(idx2, res2, acc2)
            if late is not None and fused:       # deferred form: the new step sizes and lengths, needed from step s + 2 on
                apply_fused(idx2[:0], None, dict(idx=idx2, rem=late["rem"], dt=late["dt"]), idx2[:0])
            elif late is not None:
                ts = t(idx2)
                st["dt"].index_copy_(0, ts, t(np.asarray(late["dt"], dtype=np.float64)))
                st["rem"].index_copy_(0, ts, t(np.asarray(late["rem"], dtype=np.int32)))
            if late is not None:
                dep_rem[idx2] = late["rem"]
            finish[idx2] = s + 1 + dep_rem[idx2]                     # fresh in step s + 1, already under way
            has_dep[idx2] = False
        if restart is not None and active() and not capped():
            cand = np.nonzero((finish == steps) & ~has_dep)[0]       # they complete in the step launched next
            if len(cand):
                sel, u, pn, rem = restart.predraw(cand)
                # (earlier iterations' withdrawing launches: see above.  A launch of THIS iteration is only waited for if it
                # withdrew one of the chains deposited now -- which the callers' schedules rule out, a withdrawn chain runs a
                # whole trajectory first; the check makes that an enforced invariant instead of a silent race on nxt_have)
                while withdrawn and side is not None and (withdrawn[0][0] < steps or np.isin(sel, withdrawn[0][2]).any()):
                    side.wait_event(withdrawn.pop(0)[1])
                if len(sel) and use_rec:
                    # ONE launch on the side stream (rfs_flow_deposit), its lists read from pinned memory or copied up in one piece
                    sd = side if side is not None else torch.cuda.current_stream(dev)
                    with torch.cuda.stream(sd):
                        def parts_of(put):
                            return (len(sel), put(sel, np.int32), put(u, np.float64), put(pn, np.float64),
                                    put(rem, np.int32) if rem is not None else None)
                        slot, src, offs = staged(parts_of)
                        model.flow_deposit(st, sd, src, *offs)
                        slot[1].record(sd)
                        if side is not None:
                            ev = torch.cuda.Event(); ev.record(); deposited.append(ev)
                elif len(sel):
                    # on the side stream, beside the step that is running: these chains are in mid-trajectory there and the
                    # device looks at a deposit only in the step that completes one; the next launch waits for the event
                    with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                        ts = t(np.asarray(sel, dtype=np.int64))
                        st["nxt_u"].index_copy_(0, ts, t(np.asarray(u, dtype=np.float64)))
                        st["nxt_p"].index_copy_(0, ts, t(pn))
                        if rem is not None:
                            st["nxt_rem"].index_copy_(0, ts, t(np.asarray(rem, dtype=np.int32)))
                        st["nxt_have"].index_fill_(0, ts, 1)
                        if side is not None:
                            ev = torch.cuda.Event(); ev.record(); deposited.append(ev)
                if len(sel):
                    has_dep[sel] = True
                    if rem is not None:
                        dep_rem[sel] = rem
        if dev.type == "cuda":
            ev = torch.cuda.Event(blocking=HOST_WAIT_BLOCKS); ev.record(); uploaded.append(ev)
        if not active() or capped():
            break
        if not pipeline:
            step()
    return steps

