"""-m gpu: the N > 1 code path on ONE GPU.  BASELINE configs[2] shards independent chains over the ranks of a node and
gathers the misfits on rank 0 (main_base.py:16-18,59-60,81-93); a multi-GPU node is the driver's to launch, so here
every rank of a 2-rank job sits on GPU 0 and the collectives run over gloo (RFS_BENCH_SHARED_GPU / RFS_SHARED_GPU:
RCCL refuses two ranks on one device).  Checked: the launcher starts the ranks, the gather returns the chains in
global-chain order, ranks run different chains (seed + rank), and the chains of rank 1 are exactly what a single-rank
job of rank 1's chains computes.  No scaling number comes out of this -- multi-GPU throughput stays unmeasured here."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import yaml

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(extra):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "RFS_BENCH_CHILD"):
        e.pop(k, None)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # the checks below compare the chains' state at a given DEVICE step across jobs: with hand-backs in the background a chain
    # that sits steps out is at another point of the same trajectory there, depending on when its search finished
    e["RFS_FLOW_ASYNC"] = "0"
    e.update(extra)
    return e


def test_bench_two_ranks_share_the_gpu(tmp_path):
    args = ["--chains", "256", "--steps", "2", "--warmup", "2", "--no-cpu-baseline", "--no-sampler-leg"]
    dump2 = str(tmp_path / "two.npy")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args, capture_output=True, text=True,
                       timeout=900, cwd=ROOT, env=_env({"RFS_BENCH_SHARED_GPU": "1", "RFS_BENCH_DUMP": dump2}))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["chains_per_gpu"] == 256 and d["scaling"] == "weak"
    assert "FUNCTIONAL CHECK" in d["config"]["parallelism"]
    g = np.load(dump2)
    assert g.shape == (512,) and np.isfinite(g).all() and (g > 0).all()
    assert not np.array_equal(g[:256], g[256:])                       # the ranks ran different chains (seed + rank)
    for rk in (0, 1):                                                  # ... in global-chain order: rank r's block is what
        dump1 = str(tmp_path / f"one{rk}.npy")                         # a single-rank job of rank r's chains computes
        r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--seed-rank", str(rk),
                             "--headline-only"] + args, capture_output=True, text=True, timeout=900, cwd=ROOT,
                            env=_env({"RFS_BENCH_DUMP": dump1}))
        assert r1.returncode == 0, r1.stderr[-3000:]
        assert np.array_equal(np.load(dump1), g[rk * 256:(rk + 1) * 256]), rk


def test_bench_two_ranks_on_two_gpus_over_rccl(tmp_path):
    """Where the box has two GPUs: the same two-rank job with one rank per device and backend "nccl" (RCCL) -- the process
    group, the barriers and max-over-ranks timing, and gather_misfits on DEVICE tensors.  Skipped on a one-GPU box (the
    builder's; the driver's multi-GPU node runs it)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    args = ["--chains", "256", "--steps", "2", "--warmup", "2", "--no-cpu-baseline"]
    dump2 = str(tmp_path / "two.npy")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args, capture_output=True, text=True,
                       timeout=900, cwd=ROOT, env=_env({"RFS_BENCH_DUMP": dump2}))
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["chains_per_gpu"] == 256 and d["scaling"] == "weak"
    assert "FUNCTIONAL CHECK" not in d["config"]["parallelism"]
    g = np.load(dump2)
    assert g.shape == (512,) and np.isfinite(g).all() and (g > 0).all()
    for rk in (0, 1):                                                  # rank r's block = the single-rank job of its chains
        dump1 = str(tmp_path / f"one{rk}.npy")
        r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--seed-rank", str(rk),
                             "--headline-only"] + args, capture_output=True, text=True, timeout=900, cwd=ROOT,
                            env=_env({"RFS_BENCH_DUMP": dump1}))
        assert r1.returncode == 0, r1.stderr[-3000:]
        assert np.array_equal(np.load(dump1), g[rk * 256:(rk + 1) * 256]), rk


def test_example_driver_two_ranks_share_the_gpu(tmp_path):
    """examples/main_hmc.py (the role of main_base.py) under torch.distributed.run with two ranks on GPU 0: rank 0
    broadcasts the observed data, every rank samples its own chains, rank 0 gathers [total_chains, nsamples]."""
    param = yaml.safe_load(open(os.path.join(ROOT, "examples", "param.yaml")))
    param["hmc"].update(nchains=6, nsamples=4, ndraws=1, schedule="batch")
    pfile = tmp_path / "param.yaml"
    pfile.write_text(yaml.safe_dump(param))
    out2 = tmp_path / "two"
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "examples", "main_hmc.py"), "--param", str(pfile)],
                       capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=_env({"RFS_SHARED_GPU": "1", "RFS_OUTPUT_DIR": str(out2), "OMP_NUM_THREADS": "1"}))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    m2 = np.load(out2 / "misfit.npy")
    assert m2.shape == (12, 4) and np.isfinite(m2).all()
    assert not np.array_equal(m2[:6], m2[6:])
    out1 = tmp_path / "one"
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "main_hmc.py"), "--param", str(pfile)],
                        capture_output=True, text=True, timeout=900, cwd=ROOT,
                        env=_env({"RANK": "1", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "RFS_OUTPUT_DIR": str(out1)}))
    assert r1.returncode == 0, (r1.stdout + r1.stderr)[-3000:]
    m1 = np.load(out1 / "misfit.npy")
    assert np.array_equal(m1, m2[6:])                                  # rank 1's chains, alone == inside the 2-rank job
