"""N > 1 path on CPU: world_size-2 gloo processes exercise the chain sharding and the final gather
(the path's only collective; on the GPU box the same code runs over RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rfsurfhmc_amd.chains import broadcast_setup, gather_misfits, shard_range


def test_shard_range_partitions_all_chains():
    for total in (1, 7, 8, 65536, 8191):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, total, nsamples, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dobs, x0 = (np.arange(5.0), np.arange(3.0)) if rank == 0 else (None, None)
        dobs, x0 = broadcast_setup(dobs, x0)
        assert np.array_equal(dobs, np.arange(5.0)) and np.array_equal(x0, np.arange(3.0))
        first, last = shard_range(total, rank, world)
        # "misfit" of global chain c, sample s = c + s/1000: lets rank 0 check order and content
        local = torch.arange(first, last, dtype=torch.float64)[:, None] + torch.arange(nsamples, dtype=torch.float64)[None, :] / 1000
        out = gather_misfits(local)
        if rank == 0:
            q.put(out.numpy())
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 7])
def test_two_rank_gather_matches_reference_layout(total):
    world, nsamples = 2, 5
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, nsamples, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = np.arange(total, dtype=float)[:, None] + np.arange(nsamples)[None, :] / 1000
    assert got.shape == (total, nsamples) and np.array_equal(got, want)   # misfit[ncores, nsamples], main_base.py:88
