"""CPU-only checks of the boundary and the host logic: the C-ABI library loads and exports every
symbol include/rfsurf.h declares, the product never touches the oracle, there is no CPU fallback,
and the samplers' host-side RNG / initial-model logic reproduces the reference's."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hiplib():
    from rfsurfhmc_amd import build
    build.build()               # hipcc cross-compiles for gfx950 without a GPU
    from rfsurfhmc_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol(hiplib):
    header = open(os.path.join(ROOT, "include", "rfsurf.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(rfs_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 17
    L = ctypes.CDLL(hiplib.LIBPATH)
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/rfsurf.h but not exported"
    assert declared == set(hiplib.SIGNATURES), declared ^ set(hiplib.SIGNATURES)
    hiplib.load()               # binds argtypes/restype for all of them


def test_no_cpu_fallback(hiplib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hiplib.RfsError):
        hiplib.Context(device=0)
    from rfsurfhmc_amd.model.lib import libsurf
    with pytest.raises(hiplib.RfsError):
        libsurf.forward([1.0, 0.0], [5.0, 6.0], [3.0, 3.5], [2.5, 2.7], [10.0], "Rc")
    with pytest.raises(ValueError):
        libsurf.forward([1.0, 0.0], [5.0, 6.0], [3.0, 3.5], [2.5, 2.7], [10.0], "Xx")       # bad wavetype
    with pytest.raises(ValueError):
        libsurf.forward([1.0, 0.0], [5.0, 6.0], [3.0, 3.5], [2.5, 2.7], [10.0], "Lc", mode=-1)  # bad mode
    with pytest.raises(hiplib.RfsError):                                                         # Love: device only
        libsurf.forward([1.0, 0.0], [5.0, 6.0], [3.0, 3.5], [2.5, 2.7], [10.0], "Lc", sphere=True)


def test_product_never_imports_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "rfsurfhmc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(base, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle|liboracle|oracle/_ref|hostsim", txt, flags=re.M):
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def test_chain_rng_reproduces_reference_streams(golden):
    """pyhmc/hmc.py:43,61: np.random.seed(seed + rank); chain c draws what rank c would draw."""
    from rfsurfhmc_amd.pyhmc._batched import ChainRNG, initial_models
    rng = ChainRNG(991206, 0, 3)
    for c in range(3):
        np.random.seed(991206 + c)
        assert np.random.randint(5, 21) == rng.randint([c], 5, 21)[0]
        assert np.array_equal(np.random.randn(14), rng.randn([c], 14)[0])
        assert np.random.rand() == rng.rand([c])[0]
    g = golden["sampler_hybrid"]
    x = initial_models(ChainRNG(991206, 0, 2), g["bounds"])
    assert np.array_equal(x[0], g["hmc_r0/initmodel"]) and np.array_equal(x[1], g["hmc_r1/initmodel"])
    assert np.array_equal(initial_models(ChainRNG(991206, 0, 1), g["bounds"])[0], g["da_r0/initmodel"])


def test_mirror_matches_reference_semantics():
    from rfsurfhmc_amd.pyhmc.hmcda import _mirror
    b = np.array([[0.0, 1.0], [2.0, 3.0]])
    x = np.array([[1.3, 1.5], [-0.2, 2.5], [2.7, 3.2]])       # 2.7 needs two reflections
    p = np.ones_like(x)
    xm, pm = _mirror(x, p, b)
    assert np.allclose(xm, [[0.7, 2.5], [0.2, 2.5], [0.7, 2.8]])
    assert np.array_equal(pm, [[-1, -1], [-1, 1], [1, -1]])


def test_header_is_plain_c(tmp_path):
    """include/rfsurf.h is the boundary a C / cgo / JNI binding would include: it must compile as strict C99 (and as
    C++) on its own, with no HIP or torch types in any signature."""
    import subprocess
    src = tmp_path / "use_header.c"
    src.write_text('#include "include/rfsurf.h"\n'
                   'int probe(void) { rfs_swd_params s; rfs_rf_params r; (void)s; (void)r; return (int)RFS_K_COUNT; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", f"-I{ROOT}", str(src)],
                   check=True)
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", f"-I{ROOT}", "-x", "c++", str(src)], check=True)
    code = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rfsurf.h")).read(), flags=re.S)
    assert "#include <stdint.h>" in code and code.count("#include") == 1          # no HIP / torch headers
    assert not re.search(r"hipStream_t|hipEvent_t|at::|torch::|Tensor", code)


def test_rngbatch_selftest_rejects_a_library_that_disagrees_with_numpy():
    """librngbatch.so drives numpy's private MT19937 state in place; ChainRNG only uses it after a self-test against
    numpy's own draws (rfsurfhmc_amd/pyhmc/_batched.py::_rngbatch_selftest).  A library whose numbers differ is refused."""
    from rfsurfhmc_amd.pyhmc import _batched as B
    real = B._load_rngbatch()
    if real is None:
        pytest.skip("librngbatch.so not built")
    assert B._rngbatch_selftest(real)

    class Off:                                  # same entry points, one of them off by one draw
        def __getattr__(self, name):
            return getattr(real, name)

        def rngbatch_rand(self, states, idx, n, out):
            real.rngbatch_rand(states, idx, n, out)
            real.rngbatch_rand(states, idx, n, out)
    assert not B._rngbatch_selftest(Off())
    rng = B.ChainRNG(7, 0, 2)                   # and the real one reproduces RandomState(seed + chain)
    assert np.array_equal(rng.randn([0, 1], 3), np.stack([np.random.RandomState(7 + c).randn(3) for c in range(2)]))


def test_every_option_and_statistic_the_library_knows_is_described_in_the_header():
    """rfs_set_option / rfs_stat take names; a name the source accepts and include/rfsurf.h does not mention is a setting nobody
    can find (and one whose default may change unnoticed)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "rfsurfhmc_amd", "csrc", "rfsurf_hip.hip")).read()
    hdr = open(os.path.join(root, "include", "rfsurf.h")).read()
    names = set(re.findall(r'!strcmp\(name, "([a-z_0-9]+)"\)', src)) | set(re.findall(r'!strncmp\(name, "([a-z_0-9]+)"', src))
    assert len(names) > 50
    missing = sorted(n for n in names if n not in hdr)
    assert not missing, missing
