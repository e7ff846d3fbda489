"""Test helper: oracle evaluations of many models in a process pool (the oracle is ~60-250 ms per joint evaluation;
256 models on the box's cores take a few seconds).  Test infrastructure only."""
import multiprocessing as mp
import os

import numpy as np

_O = {}


def _orc():
    if "o" not in _O:
        from oracle import oracle              # (liboracle.so was built by the parent: the `orc` fixture)
        _O["o"] = oracle
    return _O["o"]


def _ctx():
    """Workers are SPAWNED (fresh interpreters), never forked: the parent is a process with an initialised GPU runtime,
    torch and their threads, and a forked copy of it can come up holding somebody else's lock (seen: a pool that never
    returned, the third fork pool of a test session)."""
    return mp.get_context("spawn")


def _joint_worker(args):
    xs, rfpar, t, drf, dswd = args
    O = _orc()
    jo = O.Joint_RF_SWD(1.0, 1.0, O.ReceiverFunc(*rfpar), O.SurfWD(tRc=t))
    jo.set_obsdata(drf, dswd)
    ro = O.ReceiverFunc(*rfpar)
    ro.set_obsdata(drf)
    out = []
    for x in xs:
        m, g, d, f = jo.misfit_and_grad(x)
        mr, gr, dr = ro.misfit_and_grad(x)
        out.append((m, g, d, f, mr, gr, dr))
    return out


def _roots_worker(args):
    xs, t, n = args
    O = _orc()
    c = np.zeros((len(xs), len(t))); ok = np.zeros(len(xs), dtype=bool)
    for i, x in enumerate(xs):
        if not np.isfinite(x).all():             # (the restated scan never ends on a NaN model, like the reference's)
            continue
        vs, thk = x[:n], x[n:]
        vp, rho, _, _ = O.empirical_relation(vs)
        c[i], ok[i] = O.libsurf.forward(thk, vp, vs, rho, t, "Rc")
    return c, ok


def _at_roots_worker(args):
    xs, cs, rfpar, t, drf, dswd = args
    O = _orc()
    ro = O.ReceiverFunc(*rfpar)
    ro.set_obsdata(drf)
    wt = len(drf) / len(dswd)                    # model_rf_swd_vs_thk.py:79, sigma1 = sigma2
    out = []
    for x, c in zip(xs, cs):
        mr, gr, dr = ro.misfit_and_grad(x)
        ms, gs = O.swd_misfit_and_grad_at_roots(x, t, c, dswd)
        out.append(gr + wt * gs)
    return out


def joint_grad_at_roots_batch(xs, cs, rfpar, t, drf, dswd):
    """Joint gradients with the oracle's eigenfunction pass evaluated at GIVEN Rc roots cs[i] (oracle.swd_misfit_and_grad_at_roots)."""
    if len(xs) == 0:
        return np.zeros((0, xs.shape[1]))
    p = nproc()
    parts = [q for q in np.array_split(np.arange(len(xs)), p * 2) if len(q)]
    with _ctx().Pool(p) as pool:
        res = pool.map(_at_roots_worker, [(xs[q], cs[q], rfpar, t, drf, dswd) for q in parts])
    return np.array([r for part in res for r in part])


def nproc():
    return max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))


def joint_batch(xs, rfpar, t, drf, dswd):
    """[(misfit, grad, dsyn, flag, misfit_rf, grad_rf, rf)] of the oracle's joint and RF-only plugins for every row of xs."""
    p = nproc()
    parts = [q for q in np.array_split(np.arange(len(xs)), p * 2) if len(q)]
    with _ctx().Pool(p) as pool:
        res = pool.map(_joint_worker, [(xs[q], rfpar, t, drf, dswd) for q in parts])
    return [r for part in res for r in part]


def roots_pool():
    """A pool for several roots_batch calls (pass it as ``pool``)."""
    return _ctx().Pool(nproc())


def roots_batch(xs, t, n, pool=None):
    if pool is not None:
        parts = [q for q in np.array_split(np.arange(len(xs)), nproc() * 4) if len(q)]
        res = pool.map(_roots_worker, [(xs[q], t, n) for q in parts])
        return np.vstack([r[0] for r in res]), np.concatenate([r[1] for r in res])
    p = nproc()
    parts = [q for q in np.array_split(np.arange(len(xs)), p * 4) if len(q)]
    with _ctx().Pool(p) as pool:
        res = pool.map(_roots_worker, [(xs[q], t, n) for q in parts])
    return np.vstack([r[0] for r in res]), np.concatenate([r[1] for r in res])
