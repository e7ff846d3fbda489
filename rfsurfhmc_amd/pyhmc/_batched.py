"""Shared host logic of the batched samplers: per-chain legacy numpy RNG streams that reproduce
the reference's draw order (pyhmc/hmc.py:43,61: ``np.random.seed(seed + rank)``; chain c here ==
MPI rank c there), initial models, result store."""
import os

import numpy as np


class ChainRNG:
    """One legacy MT19937 stream per chain: RandomState(seed + first_chain + c) draws exactly what the
    reference's global generator draws on rank (first_chain + c)."""

    def __init__(self, seed, first_chain, nchains):
        self.rs = [np.random.RandomState(seed + first_chain + c) for c in range(nchains)]

    def rand(self, idx):
        return np.array([self.rs[c].rand() for c in idx])

    def randn(self, idx, n):
        return np.stack([self.rs[c].randn(n) for c in idx]) if len(idx) else np.zeros((0, n))

    def randint(self, idx, lo, hi):
        return np.array([self.rs[c].randint(lo, hi) for c in idx], dtype=np.int32)


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


def save_chain_results(outdir, name, rank, initmodel, obs, xmean, synmean, x_cache, syndata):
    """Result store with the reference's HDF5 member names (pyhmc/hmc.py:203-226, 272-275) as keys of
    one .npz per chain ({name}.{rank}.npz): initmodel, obs, mean/model, mean/syn, model[i], syn[i]."""
    os.makedirs(outdir, exist_ok=True)
    d = {"initmodel": initmodel, "obs": obs, "mean/model": xmean, "mean/syn": synmean, "model": x_cache}
    if syndata is not None:
        d["syn"] = syndata
    np.savez(os.path.join(outdir, f"{name}.{rank}.npz"), **d)
