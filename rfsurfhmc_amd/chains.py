"""Multi-GPU layout of the path: independent chains shard across ranks, nothing is exchanged
while sampling; the only collective on the path is the final gather of per-chain results on rank 0
(plus, when the optional ensemble mass adaptation is on, two tiny all-reduces at each adaptation point).

Mirrors the reference's whole distributed backend -- ``comm.bcast(dobs)``, ``comm.bcast(x)`` and
``comm.Gather(misfit)`` in main_base.py:59-60,90 -- with torch.distributed (backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests)."""
from typing import Tuple

import numpy as np


def shard_range(nchains_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [first, last) of global chain ids owned by ``rank`` (main_base.py:16-18:
    chain c is what the reference runs as MPI rank c, seed = seed + c)."""
    base, rem = divmod(nchains_total, world)
    first = rank * base + min(rank, rem)
    return first, first + base + (1 if rank < rem else 0)


def broadcast_setup(dobs, x0, src: int = 0):
    """main_base.py:59-60: rank 0's synthetic observed data and true model to every rank."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return dobs, x0
    objs = [dobs, x0]
    dist.broadcast_object_list(objs, src=src)
    return objs[0], objs[1]


def gather_misfits(misfit, dst: int = 0):
    """main_base.py:86-93: ``comm.Gather(tmp, misfit, root=0)`` -> [total_chains, nsamples] on rank dst.

    ``misfit``: torch tensor [local_chains, ...] (CUDA for RCCL, CPU for gloo).  Ranks may own different
    numbers of chains (remainder chains go to the first ranks), so sizes are exchanged first."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return misfit
    world, rank = dist.get_world_size(), dist.get_rank()
    n_local = torch.tensor([misfit.shape[0]], dtype=torch.int64, device=misfit.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    sizes = [int(s.item()) for s in sizes]
    nmax = max(sizes)
    pad = torch.zeros((nmax,) + tuple(misfit.shape[1:]), dtype=misfit.dtype, device=misfit.device)
    pad[: misfit.shape[0]] = misfit
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0)


def pooled_variance(x: np.ndarray) -> np.ndarray:
    """Per-parameter variance of the current models over ALL chains of the job (x: this rank's [chains, nx]).
    Two all-reduces of nx+1 doubles when a process group is up (mean first, then centred squares), plain numpy
    otherwise.  The reference has no counterpart (its `invert_Mass` is the identity, pyhmc/hmc.py:48)."""
    import torch
    import torch.distributed as dist
    x = np.asarray(x, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return x.var(axis=0)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    s1 = torch.from_numpy(np.concatenate([[float(x.shape[0])], x.sum(axis=0)])).to(dev)
    dist.all_reduce(s1)
    s1 = s1.cpu().numpy()
    n, mean = s1[0], s1[1:] / s1[0]
    s2 = torch.from_numpy(((x - mean[None, :]) ** 2).sum(axis=0)).to(dev)
    dist.all_reduce(s2)
    return s2.cpu().numpy() / n
