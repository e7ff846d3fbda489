"""Higher modes (libsurf's `mode` argument; surfdisp96.f:227-316 `do 1800 iq=1,mode`, its hand-over between modes and the
"mode not found -> zero, keep going" rule :337-362) against fixtures the COMPILED reference produced
(tests/golden/swd_modes_reference.npz, oracle/make_golden.py::gen_swd_modes): modes 1 and 2, Rayleigh and Love, flat and
spherical, on gradient / low-velocity-zone / prior / wild models and one the fundamental itself fails on.
CPU: the oracle's C restatement and the host build of the device's state machine (tests/hostsim); -m gpu: the C ABI."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim")
DP = ctypes.POINTER(ctypes.c_double); FP = ctypes.POINTER(ctypes.c_float)
P = lambda a: a.ctypes.data_as(DP)
F = lambda a: a.ctypes.data_as(FP)


def _cases(g):
    for name in sorted({k.split("/")[0] for k in g.files if k.endswith("/thk")}):
        for wt in ("Rc", "Lc"):
            for sph in (0, 1):
                for mode in (1, 2):
                    yield name, wt, sph, mode, f"{name}/{wt}/{sph}/m{mode}"


def test_oracle_restatement_of_the_mode_loop(orc, golden):
    g = golden["swd_modes_reference"]
    nzero = ntot = 0
    for name, wt, sph, mode, key in _cases(g):
        thk, vs, t = g[f"{name}/thk"], g[f"{name}/vs"], g[f"{name}/t"]
        vp, rho, _, _ = orc.empirical_relation(vs)
        c, flag = orc.libsurf.forward(thk, vp, vs, rho, t, wt, mode, bool(sph))
        assert bool(flag) == bool(g[f"{key}/fwd_flag"]), key
        assert np.array_equal(c, g[f"{key}/fwd_c"]), key                  # bit-exact, zeros of missing modes included
        nzero += int((c == 0).sum()); ntot += len(c)
        c2, ka, kb, kr, kh, flag2 = orc.libsurf.adjoint_kernel(thk, vp, vs, rho, t, wt, mode, bool(sph))
        assert bool(flag2) == bool(g[f"{key}/flag"]) and np.array_equal(c2, g[f"{key}/c"]), key
        if flag2:
            rows = g[f"{key}/rows"]
            for arr, kk in ((kb, "dcdb"), (kr, "dcdr"), (kh, "dcdh")):
                ref = g[f"{key}/{kk}"]
                if len(rows):
                    assert np.abs(arr[rows] - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-300), (key, kk)
    assert 0.05 * ntot < nzero < 0.9 * ntot          # the fixtures do contain periods at which the mode does not exist


def test_device_state_machine_runs_the_mode_loop_bit_for_bit(golden):
    """RootSearchT<.., MODES = true> (swd_math.hpp) compiled for the host: flat-earth phase velocities of modes 1 and 2
    identical to the compiled reference's, flags included (spherical results pass through _flat2sphere and are checked on
    the GPU through the ABI)."""
    so = os.path.join(HERE, "libhostsim_swd.so")
    subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-o", so, os.path.join(HERE, "hostsim_swd.cpp")], check=True)
    H = ctypes.CDLL(so)
    from oracle import oracle as O
    g = golden["swd_modes_reference"]
    n_checked = 0
    for name, wt, sph, mode, key in _cases(g):
        if sph:
            continue
        thk, vs, t = g[f"{name}/thk"], g[f"{name}/vs"], np.ascontiguousarray(g[f"{name}/t"])
        vp, rho, _, _ = O.empirical_relation(vs)
        f = [np.ascontiguousarray(np.asarray(v, dtype=np.float64).astype(np.float32)) for v in (thk, vp, vs, rho)]
        cg = np.zeros(len(t))
        fl = H.hs_rootsearch_modes(len(vs), *[F(v) for v in f], len(t), P(t), P(cg), int(wt == "Lc"), 0, mode)
        assert bool(fl) == bool(g[f"{key}/fwd_flag"]), key
        if fl:
            assert np.array_equal(cg, g[f"{key}/fwd_c"]), key
            n_checked += len(t)
    assert n_checked > 500


@pytest.mark.gpu
def test_higher_modes_through_the_abi(golden):
    """rfs_swd_forward / rfs_swd_kernel with mode = 1, 2 (libsurf.forward / adjoint_kernel of the drop-in): same flags,
    phase velocities within 1.2e-6 c of the compiled reference's (identical for almost all), zeros where the mode does
    not exist, kernels <= 2e-6 where it does -- and SurfWD(mode = 1) evaluates misfit and gradient on them."""
    from rfsurfhmc_amd.model.lib import libsurf
    from rfsurfhmc_amd.model.model_surf import SurfWD
    from oracle import oracle as O
    g = golden["swd_modes_reference"]
    nroot = nsame = 0
    for name, wt, sph, mode, key in _cases(g):
        thk, vs, t = g[f"{name}/thk"], g[f"{name}/vs"], g[f"{name}/t"]
        vp, rho, _, _ = O.empirical_relation(vs)
        c, flag = libsurf.forward(thk, vp, vs, rho, t, wt, mode, bool(sph))
        ref = g[f"{key}/fwd_c"]
        assert bool(flag) == bool(g[f"{key}/fwd_flag"]), key
        if not flag:
            continue
        assert np.array_equal(c == 0, ref == 0), key
        nz = ref != 0
        assert np.all(np.abs(c[nz] - ref[nz]) <= 1.2e-6 * ref[nz]), key
        nroot += int(nz.sum()); nsame += int((c[nz] == ref[nz]).sum())
        c2, ka, kb, kr, kh, fl2 = libsurf.adjoint_kernel(thk, vp, vs, rho, t, wt, mode, bool(sph))
        assert bool(fl2) == bool(g[f"{key}/flag"]) and np.array_equal(c2 == 0, g[f"{key}/c"] == 0), key
        rows = g[f"{key}/rows"]
        same = rows[c2[rows] == g[f"{key}/c"][rows]] if len(rows) else rows
        pos = {int(r): i for i, r in enumerate(rows)}
        for arr, kk in ((kb, "dcdb"), (kr, "dcdr"), (kh, "dcdh")):
            for r in same:
                refk = g[f"{key}/{kk}"][pos[int(r)]]
                assert np.abs(arr[r] - refk).max() <= 2e-6 * max(np.abs(refk).max(), 1e-300), (key, kk, int(r))
    assert nroot > 500 and nsame >= 0.98 * nroot, (nroot, nsame)
    # plugin level: SurfWD(mode = 1) against the oracle's plugin on the oracle's (reference-pinned) libsurf
    thk, vs, t = g["grad30/thk"], g["grad30/vs"], g["grad30/t"]
    x0 = np.hstack((vs, thk)); x1 = np.hstack((vs * 1.02, thk * 0.99))
    keep = g["grad30/Rc/0/m1/fwd_c"] != 0
    tt = t[keep][:8]                                          # periods at which the first higher mode exists
    s, o = SurfWD(mode=1, tRc=tt), O.SurfWD(mode=1, tRc=tt)
    d0, fl = o.forward(x0)
    assert fl and np.all(d0 > 0)
    s.set_obsdata(d0); o.set_obsdata(d0)
    m, gr, d, f = s.misfit_and_grad(x1)
    mo, go, do, fo = o.misfit_and_grad(x1)
    assert f and fo and np.all(np.abs(d - do) <= 1.2e-6 * do)
    assert abs(m - mo) <= 1e-4 * abs(mo) and np.abs(gr - go).max() <= 1e-4 * np.abs(go).max()
