#!/usr/bin/env python3
"""Driver with the role of the reference's main_base.py / main_DA.py on MI355X GPUs.

    python examples/main_hmc.py [--param examples/param.yaml] [--da]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/main_hmc.py

One process per GPU; every process runs `hmc.nchains` chains (global chain c is what the reference runs as MPI rank
c: seed + c, main_base.py:16-18,81).  Rank 0 makes the synthetic observed data from `true_model` and broadcasts it
(main_base.py:49-60), every rank samples, rank 0 gathers the misfits over RCCL and writes `misfit.npy`
[total_chains, nsamples] and `real_syn.npy` (main_base.py:56,86-93).  Per-rank results: `{name}.rank{r}.h5`
(`.npz` where no HDF5 library exists); per-chain `{name}.{chain}.h5` as the reference writes them for <= 16 chains."""
import argparse
import os
import sys
import time

import numpy as np
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def bounds_from_true_model(vs, thk):
    """The search range of main_base.py:64-77: vs +-80 % clipped to [1.5, 5], thickness +-20 %, last thickness [0, 2]."""
    n = len(vs)
    b = np.ones((2 * n, 2))
    b[:n, 0] = np.maximum(vs - vs * 0.8, 1.5); b[:n, 1] = np.minimum(vs + vs * 0.8, 5.0)
    b[n:, 0] = thk - thk * 0.2; b[n:, 1] = thk + thk * 0.2
    b[-1, :] = 0.0, 2.0
    return b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--param", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "param.yaml"))
    ap.add_argument("--da", action="store_true", help="dual-averaging sampler (main_DA.py) instead of plain HMC")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from rfsurfhmc_amd.chains import broadcast_setup, gather_misfits
    from rfsurfhmc_amd.model.model_rf import ReceiverFunc
    from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
    from rfsurfhmc_amd.model.model_surf import SurfWD
    from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC
    from rfsurfhmc_amd.pyhmc.hmcda import HMCDualAveraging

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # RFS_SHARED_GPU=1: functional check of a multi-rank job on a box with ONE GPU (every rank on device 0, collectives
    # over gloo: RCCL refuses two ranks on one device)
    shared = os.environ.get("RFS_SHARED_GPU") == "1"
    if shared:
        local = 0
    torch.cuda.set_device(local)                          # before the process group: RCCL binds to the current device
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo" if shared else "nccl")          # "nccl" is RCCL

    with open(args.param) as f:
        param = yaml.safe_load(f)
    model_swd = SurfWD.init(**param["swd"]); model_swd.device = local
    model_rf = ReceiverFunc.init(**param["rf"]); model_rf.device = local
    thk = np.asarray(param["true_model"]["thk"], dtype=float)
    vs = np.asarray(param["true_model"]["vs"], dtype=float)
    model_swd.set_thk(thk); model_rf.set_thk(thk)
    model = Joint_RF_SWD(1.0, 1.0, model_rf, model_swd, device=local)
    hp = dict(param["hmc"])
    if os.environ.get("RFS_OUTPUT_DIR"):
        hp["OUTPUT_DIR"] = os.environ["RFS_OUTPUT_DIR"]
    outdir = hp["OUTPUT_DIR"]
    os.makedirs(outdir, exist_ok=True)

    dobs = x = None
    if rank == 0 or world == 1:                           # (world == 1 with RANK = r: the chains of rank r, alone)
        x = np.hstack((vs, thk))
        drsyn, dssyn, _ = model.forward(x)
        dobs = np.concatenate((drsyn, dssyn))
        np.save(os.path.join(outdir, "real_syn.npy"), dobs)
    dobs, x = broadcast_setup(dobs, x)
    nt = model.rfmodel.nt
    model.set_obsdata(dobs[:nt], dobs[nt:])
    boundaries = bounds_from_true_model(vs, thk)

    schedule = hp.pop("schedule", "batch")
    cls = HMCDualAveraging if args.da else HamitonianMC
    chain = cls.init(model, boundaries, rank, **hp)
    chain.verbose = False
    t0 = time.time()
    tmp = chain.sample_flow() if schedule == "flow" else chain.sample()
    tmp = np.atleast_2d(tmp)
    el = time.time() - t0
    tmp_t = torch.from_numpy(np.ascontiguousarray(tmp))
    misfit = gather_misfits(tmp_t if shared else tmp_t.cuda())
    if rank == 0 or world == 1:
        misfit = misfit.cpu().numpy()
        np.save(os.path.join(outdir, "misfit.npy"), misfit)
        print(f"{misfit.shape[0]} chains x {misfit.shape[1]} samples on {world} GPU(s) in {el:.1f} s; "
              f"median final misfit {np.median(misfit[:, -1]):.4g}; results in {outdir}")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
