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


def _canned_result():
    """A result dictionary as rich as round 4's 21 KB line (profiles/r04_bench_n1.json), grown further."""
    import bench
    d = json.loads(open(os.path.join(ROOT, "profiles", "r04_bench_n1.json")).read().strip().splitlines()[-1])
    d["config3"]["padding"] = ["x" * 100] * 200                       # legs may grow without the line growing
    d["config"]["set_up_note"] = "n" * 5000
    d["cpu_baseline"]["sample"] = "s" * 3000
    d["sustained_value"], d["sustained_ms_per_step"], d["sustained_steps"] = 1.5e6, 5.4, 100
    return bench, d


def test_headline_line_is_short_and_round_trips():
    """the driver keeps 8 KB of stdout; the line must fit with margin whatever the legs carry."""
    bench, d = _canned_result()
    line = bench.headline_line(d)
    assert len(line) < 6000 and "\n" not in line
    out = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["dtype"] == d["dtype"] and out["value"] == pytest.approx(d["value"], rel=1e-6)
    assert out["config"]["workload"].startswith("configs[1]") and "model" not in out["config"]
    for k in ("chains_per_gpu", "nlayer", "nt", "nper", "dt", "accept_ratio", "root_search_mode", "set_up_steps"):
        assert k in out["config"], k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in out["roofline"], k
    assert out["roofline"]["frac"] == pytest.approx(out["roofline"]["achieved"] / out["roofline"]["peak"], rel=1e-5)
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in out["cpu_baseline"], k
    assert out["sustained_value"] == 1.5e6 and out["config3_value"] == pytest.approx(d["config3_value"], rel=1e-6)
    assert not any(isinstance(v, (dict, list)) for k, v in out.items() if k not in ("config", "roofline", "cpu_baseline"))


def test_emit_prints_the_line_last_and_writes_the_detail_file(tmp_path, monkeypatch, capsys):
    bench, d = _canned_result()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit(dict(d, _cpu_inputs={"xs": [1.0] * 10000}))
    out = capsys.readouterr().out
    last = out.rstrip("\n").splitlines()[-1]
    assert json.loads(last)["detail"] == bench.DETAIL_FILE and len(last) < 6000
    full = json.load(open(tmp_path / bench.DETAIL_FILE))
    assert "dt_sweep" in full and "config3" in full and "_cpu_inputs" not in full
