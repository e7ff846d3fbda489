"""`python bench.py --gpus N` must start N ranks by itself (the reference's `mpiexec -n N python main_base.py`,
main_base.py:16-18,59-60,90).  Here, without a GPU, the same launcher runs its ranks with --dry-run: gloo process
group, barrier, max-over-ranks time, the final gather through rfsurfhmc_amd.chains -- no device work."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "RFS_BENCH_CHILD"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)


@pytest.mark.parametrize("n", [1, 2])
def test_gpus_flag_launches_that_many_ranks(n):
    r = _run(["--gpus", str(n), "--steps", "4", "--warmup", "1", "--chains", "7", "--dry-run"])
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                     # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["gathered_chains"] == 7 * n and d["dry_run"] is True
    assert d["steps"] == 4 and abs(d["ms_per_step"] - n) < 1e-9     # max over ranks of the fake times (rank r: r+1 ms)


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "2", "--dry-run"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_launching_process_never_touches_the_gpu():
    """The parent of the ranks must stay clean: no torch.cuda call, no librfsurf_hip load before the children exist."""
    import ast
    tree = ast.parse(open(BENCH).read())
    fn = {f.name: f for f in tree.body if isinstance(f, ast.FunctionDef)}
    bad = {"torch", "rfsurfhmc_amd", "_lib"}
    for name in ("launch", "main", "cpu_baseline", "_host_cpus"):
        for node in ast.walk(fn[name]):
            if isinstance(node, ast.Import):
                assert not any(a.name.split(".")[0] in bad for a in node.names), (name, ast.dump(node))
            if isinstance(node, ast.ImportFrom):
                assert (node.module or "").split(".")[0] not in bad, (name, ast.dump(node))
            if isinstance(node, ast.Name):
                assert node.id not in bad, (name, node.id)
