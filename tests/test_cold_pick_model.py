"""The selection rule of the search without a prediction (k_swd_cold_scan / k_swd_cold_pick, round 6), as a host model: every
period's secular function on ONE grid (start value + i dc), every sign change refined, and getsol's scan (surfdisp96.f:433-479)
replayed on the refined roots -- from the root before - 1.5 dc, upwards if the sign there is the sign below every root, else
downwards, to the first step of dc with an odd number of roots in it -- against the oracle's sequential search on wild models.
The device kernels follow this rule with lanes for loops; what the rule cannot see (a pair of roots inside one table cell that the
reference's grid, which hangs on the root before, happens to split) is what the branch test on the device is there for."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
from scipy.optimize import brentq

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim")
DP = ctypes.POINTER(ctypes.c_double)
FP = ctypes.POINTER(ctypes.c_float)
P = lambda a: a.ctypes.data_as(DP)
F = lambda a: a.ctypes.data_as(FP)
DC = float(np.float32(0.005))


@pytest.fixture(scope="module")
def hs():
    so = os.path.join(HERE, "libhostsim_swd.so")
    subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-std=c++17"] + os.environ.get("RFS_HOSTSIM_CXXFLAGS", "").split() +
                   ["-o", so, os.path.join(HERE, "hostsim_swd.cpp")], check=True)
    lib = ctypes.CDLL(so)
    lib.hs_start_value.restype = ctypes.c_double
    lib.hs_secular.restype = ctypes.c_double
    return lib


def replay(roots, s0, cc, bmx):
    """roots[k]: ascending refined roots of period k; s0[k]: sign bit of the function at the start value.  Returns the picked root per
    period, or None where the rule has no answer (getsol's clamp at the start value, no root below the fastest layer + dc)."""
    out, rprev = [], 0.0
    for k, R in enumerate(roots):
        c1 = cc if k == 0 else rprev - 1.5 * DC
        if k > 0 and not c1 > cc:
            return out + [None] * (len(roots) - k)
        nb = int(np.sum(R < c1))
        up = k == 0 or ((s0[k] ^ (nb & 1)) == s0[0])
        seq = R[nb:] if up else R[:nb][::-1]
        d = (seq - c1) if up else (c1 - seq)
        pick, i = None, 0
        while i < len(seq):
            m = np.ceil(d[i] / DC)
            if up and c1 + (m - 1) * DC >= bmx + DC:
                break
            if not up and c1 - m * DC <= cc:
                break
            cnt = int(np.sum(d[i:] <= m * DC))
            if cnt % 2:
                pick = float(seq[i])
                break
            i += cnt
        if pick is None or pick > bmx:
            return out + [None] * (len(roots) - k)
        out.append(pick)
        rprev = pick
    return out


@pytest.mark.parametrize("wave,love", [("Rc", 0), ("Lc", 1)])
def test_replay_of_the_scan_on_refined_roots_picks_the_sequential_searchs_roots(hs, orc, wave, love):
    thk0 = np.array([3., 3, 4, 5, 5, 6, 7, 8, 10, 0]); vs0 = np.linspace(2.9, 4.6, 10)
    t = np.ascontiguousarray(np.arange(5., 41.))
    x0 = np.hstack((vs0, thk0))
    lo, hi = 0.5 * x0, 1.5 * x0                       # (wide bounds: velocity inversions, thin and thick layers)
    L = orc._LibSurf()
    rng = np.random.default_rng(21 + love)
    nper_all = nsame = nmodels = nfull = nanom = 0
    for it in range(60):
        x = lo + (hi - lo) * rng.random(20); x[19] = 0.0
        vs, thk = x[:10], x[10:]
        vp, rho, _, _ = orc.empirical_relation(vs)
        cref, ok = L.forward(thk, vp, vs, rho, t, wave)
        if not ok:
            continue
        f = [np.ascontiguousarray(a.astype(np.float32)) for a in (thk, vp, vs, rho)]
        bmx = ctypes.c_float(0)
        cc = hs.hs_start_value(10, *[F(a) for a in f], ctypes.byref(bmx))
        bmx = float(bmx.value)
        npts = int(np.floor((bmx + DC - cc) / DC)) + 2
        tab = np.zeros((len(t), npts))
        hs.hs_secular_table(10, *[F(a) for a in f], len(t), P(t), npts, ctypes.c_double(cc), ctypes.c_double(DC), love, P(tab))
        roots, s0 = [], []
        for k in range(len(t)):
            om = 2.0 * np.pi / t[k]
            fun = lambda c: hs.hs_secular(10, *[F(a) for a in f], ctypes.c_double(om), ctypes.c_double(c), love)
            sg = np.signbit(tab[k])
            cells = np.nonzero(sg[1:] != sg[:-1])[0]
            roots.append(np.array([brentq(fun, cc + i * DC, cc + (i + 1) * DC, xtol=1e-12) for i in cells]))
            s0.append(int(sg[0]))
        pick = replay(roots, s0, cc, bmx)
        nmodels += 1
        good = [p is not None and abs(p - c) <= 1.2e-6 * c for p, c in zip(pick, cref)]
        nper_all += len(t); nsame += int(np.sum(good)); nfull += int(all(good))
        nanom += int(np.any(np.diff(cref) <= -1.5 * DC))
    print(f"{wave}: {nmodels} wild models ({nanom} with anomalous dispersion somewhere): the replay picks the sequential search's root for "
          f"{nsame} of {nper_all} periods, every period of {nfull} models")
    # (Love: the fundamental mode's dispersion is normal on all of these models; its three misses in 2 160 are pairs of roots 2 m/s
    # apart inside ONE table cell -- e.g. 4.12852 and 4.13058 km/s at T = 28 s on a model with three low-velocity layers -- which the
    # reference's grid happens to split and the table cannot see: the case the branch test on the device is there for)
    assert nmodels >= 25 and (love or nanom >= 5)
    assert nsame >= 0.97 * nper_all and nfull >= 0.85 * nmodels
