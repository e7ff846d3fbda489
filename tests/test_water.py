"""Models with a water layer on top (vs(1) = 0): surfdisp96's water branch (surfdisp96.f:138-139 llw, :149-160 / :201-206
start value from the water's P velocity, :870-886 the fluid layer in dltar4, :750 Love stops at the sea floor) and the fluid
branches of sregn96 (varsv :858-877, dnka :555-575, hska :931-945, evalg :778-811, intijr :1245-1262, energy :1122-1140,
getdcdh :1456-1509, getmat :1551-1565) against fixtures the COMPILED reference produced
(tests/golden/swd_water_reference.npz, oracle/make_golden.py::gen_swd_water): shelf to deep ocean, 6 to 30 layers.
The C restatement under oracle/ does not cover water layers -- these tests stand on the reference's own output.
CPU: the device's lane math built for the host (tests/hostsim); -m gpu: the C ABI (libsurf drop-in)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim")
DP = ctypes.POINTER(ctypes.c_double); FP = ctypes.POINTER(ctypes.c_float)
P = lambda a: a.ctypes.data_as(DP)
F = lambda a: a.ctypes.data_as(FP)
NAMES = ("shelf_0p2", "ocean_2", "ocean_4p5", "ocean_lvz", "ocean_30", "sediment")


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


def _model(g, name):
    return tuple(np.ascontiguousarray(g[f"{name}/{k}"]) for k in ("thk", "vp", "vs", "rho", "t"))


def test_fixture_is_what_it_says(golden):
    g = golden["swd_water_reference"]
    for name in NAMES:
        thk, vp, vs, rho, t = _model(g, name)
        assert vs[0] == 0.0 and np.all(vs[1:] > 0) and thk[-1] == 0.0
        for wt in ("Rc", "Rg", "Lc"):
            for sph in (0, 1):
                assert bool(g[f"{name}/{wt}/{sph}/m0/fwd_flag"]) and np.all(np.isfinite(g[f"{name}/{wt}/{sph}/m0/fwd_c"]))
        assert np.all(np.isnan(g[f"{name}/Lg/0/m0/fwd_c"]))          # the reference's slegn96 on a water model


def test_device_lane_math_on_the_host(golden):
    """swd_secular / swd_secular_love with the water branch under the reference-semantics search: flat-earth phase
    velocities bit-identical to the compiled reference's; sr_up / sr_down_energy<WATER> kernels (flat and spherical) to
    1e-9 (observed 1e-13)."""
    so = os.path.join(HERE, "libhostsim_swd.so")
    subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-o", so, os.path.join(HERE, "hostsim_swd.cpp")], check=True)
    H = ctypes.CDLL(so)
    H.hs_eigen_general.restype = ctypes.c_double
    g = golden["swd_water_reference"]
    nroots = 0
    for name in NAMES:
        thk, vp, vs, rho, t = _model(g, name)
        n = len(vs)
        f = [np.ascontiguousarray(v.astype(np.float32)) for v in (thk, vp, vs, rho)]
        for wt in ("Rc", "Lc"):
            cg = np.zeros(len(t))
            fl = H.hs_rootsearch_general(n, *[F(v) for v in f], len(t), P(t), P(cg), int(wt == "Lc"), 0)
            assert fl == 1 and np.array_equal(cg, g[f"{name}/{wt}/0/m0/fwd_c"]), (name, wt)
            nroots += len(t)
        for sph in (0, 1):
            key = f"{name}/Rc/{sph}/m0"
            cflat = np.zeros(len(t))
            assert H.hs_rootsearch_general(n, *[F(v) for v in f], len(t), P(t), P(cflat), 0, sph) == 1
            for k in range(len(t)):
                da, db, dh, dr = (np.zeros(n) for _ in range(4))
                cp = ctypes.c_double(cflat[k])
                H.hs_eigen_general(n, *[F(v) for v in f], ctypes.c_double(t[k]), ctypes.byref(cp), P(da), P(db), P(dh), P(dr), 0, sph)
                assert abs(cp.value - g[f"{key}/c"][k]) <= 1e-12 * cp.value, (key, k)
                assert db[0] == 0.0
                for arr, ref in ((da, g[f"{key}/dcda"][k]), (db[1:], g[f"{key}/dcdb_solid"][k]), (dr, g[f"{key}/dcdr"][k]),
                                 (dh, g[f"{key}/dcdh"][k])):
                    assert rel(arr, ref) < 1e-9, (key, k, rel(arr, ref))
    assert nroots > 150


@pytest.mark.gpu
def test_water_layer_models_through_the_abi(golden):
    """libsurf.forward / adjoint_kernel of the drop-in on the fixture models: flags, phase velocities (identical for
    almost all, all within 1.2e-6 c), group velocities, first higher mode, flat and spherical; Rayleigh kernels <= 2e-6
    (Rc) / 2e-5 (Rg); the water layer's dcdb is 0; Love kernels and Lg values are NaN as the reference's are."""
    from rfsurfhmc_amd.model.lib import libsurf
    g = golden["swd_water_reference"]
    nroot = nsame = 0
    for name in NAMES:
        thk, vp, vs, rho, t = _model(g, name)
        for wt in ("Rc", "Rg", "Lc", "Lg"):
            for sph in (0, 1):
                for mode in ((0, 1) if wt[1] == "c" else (0,)):
                    key = f"{name}/{wt}/{sph}/m{mode}"
                    c, flag = libsurf.forward(thk, vp, vs, rho, t, wt, mode, bool(sph))
                    ref = g[f"{key}/fwd_c"]
                    assert bool(flag) == bool(g[f"{key}/fwd_flag"]), key
                    if wt == "Lg":
                        assert np.all(np.isnan(c)) and np.all(np.isnan(ref)), key
                        continue
                    assert np.array_equal(c == 0, ref == 0), key
                    nz = ref != 0
                    tol = 1.2e-6 if wt[1] == "c" else 2e-5
                    assert np.all(np.abs(c[nz] - ref[nz]) <= tol * np.abs(ref[nz])), (key, np.abs(c[nz] / ref[nz] - 1).max())
                    if wt[1] == "c":
                        nroot += int(nz.sum()); nsame += int((c[nz] == ref[nz]).sum())
                    if wt[0] == "R" and mode == 0:
                        c2, ka, kb, kr, kh, fl2 = libsurf.adjoint_kernel(thk, vp, vs, rho, t, wt, mode, bool(sph))
                        assert fl2 and np.all(np.abs(c2 - g[f"{key}/c"]) <= tol * np.abs(c2)), key
                        assert not np.any(kb[:, 0]), key
                        ktol = 2e-6 if wt == "Rc" else 2e-5
                        same = np.nonzero(c2 == g[f"{key}/c"])[0] if wt == "Rc" else np.arange(len(t))
                        for arr, kk in ((ka, "dcda"), (kb[:, 1:], "dcdb_solid"), (kr, "dcdr"), (kh, "dcdh")):
                            for r in same:
                                assert rel(arr[r], g[f"{key}/{kk}"][r]) <= ktol, (key, kk, int(r), rel(arr[r], g[f"{key}/{kk}"][r]))
        c2, ka, kb, kr, kh, fl2 = libsurf.adjoint_kernel(thk, vp, vs, rho, t, "Lc")
        assert fl2 and np.all(np.abs(c2 - g[f"{name}/Lc/0/m0/fwd_c"]) <= 1.2e-6 * c2) and np.all(np.isnan(kb[:, 1:]))
    assert nroot > 400 and nsame >= 0.98 * nroot, (nroot, nsame)


@pytest.mark.gpu
def test_water_and_solid_models_in_one_batch(golden):
    """A batch that mixes models with and without a water layer: every chain's result equals its own single call."""
    from rfsurfhmc_amd.model.lib import libsurf
    from oracle import oracle as O
    g = golden["swd_water_reference"]
    thk, vp, vs, rho, t = _model(g, "ocean_2")
    n = len(vs)
    vs2 = np.linspace(3.0, 4.7, n); thk2 = np.r_[np.full(n - 1, 5.0), 0.0]
    vp2, rho2, _, _ = O.empirical_relation(vs2)
    T, A, B, R = (np.vstack(p) for p in ((thk, thk2, thk), (vp, vp2, vp), (vs, vs2, vs), (rho, rho2, rho)))
    cb, kab, kbb, krb, khb, fb = libsurf.adjoint_kernel(T, A, B, R, t, "Rc")
    assert np.all(fb)
    c0, ka0, kb0, kr0, kh0, f0 = libsurf.adjoint_kernel(thk, vp, vs, rho, t, "Rc")
    assert np.array_equal(cb[0], c0) and np.array_equal(cb[2], c0) and np.array_equal(khb[0], kh0) and np.array_equal(kab[2], ka0)
    c1, ka1, kb1, kr1, kh1, f1 = libsurf.adjoint_kernel(thk2, vp2, vs2, rho2, t, "Rc")
    co, kao, kbo, kro, kho, fo = O.libsurf.adjoint_kernel(thk2, vp2, vs2, rho2, t, "Rc")
    assert np.all(np.abs(cb[1] - co) <= 1.2e-6 * co) and np.all(np.abs(c1 - co) <= 1.2e-6 * co)
    same = np.nonzero(cb[1] == co)[0]
    assert len(same) >= len(t) - 1
    for r in same:
        assert rel(kbb[1][r], kbo[r]) < 2e-6 and rel(khb[1][r], kho[r]) < 2e-6
