"""The reference's on-disk format, {name}.{rank}.h5 (pyhmc/hmc.py:58,203-226,272-275; read back by
src/plot_results.py:106-156), against files the UNMODIFIED reference samplers wrote through a real h5py:
tests/golden/reference_store/{hmc,da}.0.h5 (oracle/make_golden.py --h5-store, run under the image's conda python3.9,
the one interpreter here that has h5py; same seeded runs as sampler_hybrid.npz's hmc_r0 / da_r0).

CPU part: the product's HDF5 writer (h5py where importable, else libhdf5 through ctypes) reproduces those files member
for member -- names, shapes, types, values -- and a real h5py, where one exists, reads the product's file the way
plot_results.py does.  The GPU part (the product's own sampler run landing in the same file) is in
test_gpu_samplers.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
STORE = os.path.join(HERE, "golden", "reference_store")
H5PY_PYTHON = os.environ.get("RFS_H5PY_PYTHON", "/opt/conda/bin/python3.9")


def _need_backend():
    from rfsurfhmc_amd.pyhmc import _h5
    if _h5.backend() is None:
        pytest.skip("neither h5py nor libhdf5 here")
    return _h5


def _real_h5py():
    ok = os.path.exists(H5PY_PYTHON) and subprocess.run([H5PY_PYTHON, "-c", "import h5py"], capture_output=True).returncode == 0
    if not ok:
        pytest.skip(f"no interpreter with a real h5py ({H5PY_PYTHON})")
    return H5PY_PYTHON


def _listing(h5, f):
    return {name: (tuple(d.shape), np.dtype(d.dtype).str, np.asarray(d[...])) for name, d in h5.walk(f)}


@pytest.mark.parametrize("tag", ["hmc", "da"])
def test_reference_written_files_are_read_and_reproduced(tag, tmp_path):
    h5 = _need_backend()
    from rfsurfhmc_amd.pyhmc._batched import load_chain_results, save_chain_results
    ref_path = os.path.join(STORE, f"{tag}.0.h5")
    d = load_chain_results(ref_path)
    g = np.load(os.path.join(HERE, "golden", "sampler_hybrid.npz"))
    key = {"hmc": "hmc_r0", "da": "da_r0"}[tag]
    # the file is the seeded run the sampler traces were recorded from (two interpreters: agreement to rounding)
    assert d["model"].shape == (6, 14) and d["syn"].shape == (6, 197)
    np.testing.assert_allclose(d["initmodel"], g[key + "/initmodel"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(d["obs"], g["dobs"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(d["model"][-1], g[key + "/x"][-1], rtol=0, atol=1e-12)
    # write the same content with the product's writer: member for member the reference's file
    mine = save_chain_results(str(tmp_path), tag, 0, d["initmodel"], d["obs"], d["mean/model"], d["mean/syn"],
                              d["model"], d["syn"], fmt="h5")
    assert os.path.basename(mine) == f"{tag}.0.h5"
    with h5.open_file(ref_path, "r") as fr, h5.open_file(mine, "r") as fm:
        assert sorted(fr.keys()) == sorted(fm.keys()) == sorted(["initmodel", "obs", "mean"] + [str(i) for i in range(6)])
        a, b = _listing(h5, fr), _listing(h5, fm)
    assert set(a) == set(b) == {"initmodel", "obs", "mean/model", "mean/syn"} | {f"{i}/{k}" for i in range(6) for k in ("model", "syn")}
    for name in a:
        assert a[name][0] == b[name][0] and a[name][1] == b[name][1] == "<f8", name
        assert np.array_equal(a[name][2], b[name][2]), name


def test_a_real_h5py_reads_the_products_file_like_plot_results(tmp_path):
    """h5py (the reference's reader) on a file written through libhdf5/ctypes: the lookups of
    src/plot_results.py:117-146 -- fio["initmodel"][:], fio["mean/model"][:], fio[f"{i}/syn"][:] ..."""
    _need_backend()
    py = _real_h5py()
    from rfsurfhmc_amd.pyhmc._batched import load_chain_results, save_batched_results, save_chain_results
    rng = np.random.default_rng(7)
    init, obs, xm, sm = rng.random(14), rng.random(197), rng.random(14), rng.random(197)
    xs, syn = rng.random((5, 14)), rng.random((5, 197))
    p = save_chain_results(str(tmp_path), "mine", 3, init, obs, xm, sm, xs, syn, fmt="h5")
    pb = save_batched_results(str(tmp_path), "mine", 0, 3, init[None], obs, xm[None], sm[None], xs[None], syn[None],
                              rng.random((1, 5)), fmt="h5")
    script = (
        "import h5py, json, sys, numpy as np\n"
        "f = h5py.File(sys.argv[1], 'r'); out = {}\n"
        "f.visititems(lambda n, o: out.__setitem__(n, [list(o.shape), o.dtype.str, np.asarray(o[()]).ravel().tolist()])"
        " if isinstance(o, h5py.Dataset) else None)\n"
        "out['_lookups'] = [f['initmodel'][:].tolist(), f['mean/model'][:].tolist(), f['4/syn'][:].tolist()]"
        " if 'mean' in f else []\n"
        "print(json.dumps(out))\n")
    got = json.loads(subprocess.run([py, "-c", script, p], capture_output=True, text=True, check=True).stdout)
    look = got.pop("_lookups")
    assert look[0] == init.tolist() and look[1] == xm.tolist() and look[2] == syn[4].tolist()
    want = {"initmodel": init, "obs": obs, "mean/model": xm, "mean/syn": sm}
    want.update({f"{i}/model": xs[i] for i in range(5)}); want.update({f"{i}/syn": syn[i] for i in range(5)})
    assert set(got) == set(want)
    for k, v in want.items():
        assert got[k][0] == list(v.shape) and got[k][1] == "<f8" and got[k][2] == v.ravel().tolist(), k
    # the batched per-rank file as well
    got = json.loads(subprocess.run([py, "-c", script, pb], capture_output=True, text=True, check=True).stdout)
    got.pop("_lookups")
    assert got["model"][0] == [1, 5, 14] and got["model"][2] == xs.ravel().tolist() and got["first_chain"][2] == [3]
    assert got["first_chain"][1] == "<i8"
    # and back through the product's reader
    d = load_chain_results(p)
    assert np.array_equal(d["model"], xs) and np.array_equal(d["syn"], syn) and np.array_equal(d["mean/syn"], sm)


def test_binding_surface_used_by_the_reference_writer(tmp_path):
    """The calls of pyhmc/hmc.py:203-226: create_group, create_dataset(name, dtype='f8', shape=...), fio[name][:] = x,
    create_dataset(name, data=...)."""
    h5 = _need_backend()
    if h5.backend() != "libhdf5":
        pytest.skip("h5py present: the ctypes binding is not what open_file hands out")
    path = str(tmp_path / "w.h5")
    x, syn = np.linspace(0, 1, 14), np.linspace(2, 3, 197)
    fio = h5.File(path, "w")
    fio.create_dataset("initmodel", data=x)
    fio.create_group("7")
    fio.create_dataset("7/model", dtype="f8", shape=x.shape)
    fio["7/model"][:] = x
    fio.create_dataset("7/syn", dtype="f8", shape=syn.shape)
    fio["7/syn"][:] = syn[:]
    fio.create_dataset("deep/er/flags", data=np.array([True, False]))
    fio.create_dataset("count", data=np.int32(4))
    fio.close()
    with h5.File(path) as f:
        assert sorted(f.keys()) == ["7", "count", "deep", "initmodel"] and "7/syn" in f and "7/nope" not in f
        assert np.array_equal(f["7/model"][:], x) and np.array_equal(f["7/syn"][3:5], syn[3:5])
        assert f["deep/er/flags"].dtype == np.uint8 and f["deep/er/flags"][:].tolist() == [1, 0]
        assert f["count"].shape == () and int(f["count"][()]) == 4 and f["count"].dtype == np.int32
        assert len(f["7"]) == 2 and list(f["deep"]) == ["er"]
        with pytest.raises(KeyError):
            f["missing"]
        with pytest.raises(OSError):
            f.create_dataset("x", data=x)                 # read-only file
    with h5.File(path, "r+") as f:
        f["7/model"][...] = 2 * x
        with pytest.raises(OSError):
            f.create_dataset("initmodel", data=x)         # exists already
        with pytest.raises(TypeError):
            f.create_dataset("c", data=np.zeros(2, dtype=complex))
    with h5.File(path) as f:
        assert np.array_equal(f["7/model"][:], 2 * x)
    with pytest.raises(OSError):
        h5.File(str(tmp_path / "absent.h5"), "r")


def test_the_references_plot_script_runs_on_the_products_files(tmp_path):
    """plot.py + src/plot_results.py of the reference, unmodified, under the interpreter that has h5py and matplotlib,
    on an output directory written by the product ({name}.{chain}.h5, misfit.npy, real_syn.npy -- main_base.py:56,93):
    all six figures come out.  Runs only where /root/reference exists (this container)."""
    _need_backend()
    py = _real_h5py()
    ref = "/root/reference"
    if not os.path.exists(os.path.join(ref, "plot.py")):
        pytest.skip("reference tree not present")
    if subprocess.run([py, "-c", "import matplotlib, yaml"], capture_output=True).returncode != 0:
        pytest.skip("no matplotlib / yaml beside h5py")
    import yaml
    from rfsurfhmc_amd.pyhmc._batched import load_chain_results, save_chain_results
    d = load_chain_results(os.path.join(STORE, "hmc.0.h5"))
    mis = np.load(os.path.join(STORE, "hmc.misfit.npy"))
    par = yaml.safe_load(open(os.path.join(ref, "param.yaml")))
    par["hmc"]["OUTPUT_DIR"], par["hmc"]["name"] = "./results/", "mine"
    yaml.safe_dump(par, open(tmp_path / "param.yaml", "w"))
    out = str(tmp_path / "results")
    for c in range(2):                                    # two chains = two of the reference's MPI ranks
        f = 1 + 0.01 * c
        save_chain_results(out, "mine", c, d["initmodel"] * f, d["obs"], d["mean/model"] * f, d["mean/syn"],
                           d["model"] * f, d["syn"], fmt="h5")
    np.save(os.path.join(out, "misfit.npy"), np.stack([mis, 1.1 * mis]))
    np.save(os.path.join(out, "real_syn.npy"), d["obs"])
    r = subprocess.run([py, os.path.join(ref, "plot.py")], cwd=str(tmp_path), capture_output=True, text=True,
                       env=dict(os.environ, MPLBACKEND="Agg"))
    assert r.returncode == 0, r.stderr[-2000:]
    figs = {"best_model.png", "best_syn_fit.png", "best_model_hist.png", "misfit.png", "plot_best_fit_hist.png", "static.png"}
    assert figs <= set(os.listdir(out)), sorted(os.listdir(out))
    assert all(os.path.getsize(os.path.join(out, f)) > 10_000 for f in figs)
