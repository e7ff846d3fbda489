"""Drop-in for the reference's pybind11 extension ``libsurf`` (src/SWD/main.cpp:84-94):
same function names, argument order, defaults and return tuples, executed by the HIP
library.  Additionally accepts 2-D model arrays [nchain, nlayer] (then every output gains a
leading chain axis and the flag becomes a bool array).

All four wavetypes (Rc, Rg, Lc, Lg), flat or spherical earth, fundamental and higher modes
(``mode`` = 0, 1, 2, ...: the reference's mode loop, surfdisp96.f:227-316).

Differences that are deliberate: a bad ``wavetype`` raises ValueError instead of calling
exit(0) (main.cpp:24); the Love kernels return dcda = 0 where the reference returns an uninitialised array
(surfdisp.cpp:258-296 never writes it)."""
import numpy as np

from ..._lib import RFS_WAVE, RfsError, default_context, hptr

_WAVES = ("Rc", "Rg", "Lc", "Lg")


def _prep(thk, vp, vs, rho, period):
    arrs = [np.ascontiguousarray(np.asarray(a, dtype=np.float64)) for a in (thk, vp, vs, rho)]
    single = arrs[0].ndim == 1
    arrs = [np.atleast_2d(a) for a in arrs]
    t = np.ascontiguousarray(np.asarray(period, dtype=np.float64)).ravel()
    return arrs, t, single


def _check(wavetype, mode, sphere):
    if wavetype not in _WAVES:
        raise ValueError("wavetype should be one of [Rc,Rg,Lc,Lg]")
    if int(mode) != mode or mode < 0:
        raise ValueError("mode should be a non-negative integer (0 = fundamental)")


def forward(thk, vp, vs, rho, period, wavetype, mode=0, sphere=False, device=0):
    """(c[nt] f64, bool) -- libsurf.forward, src/SWD/main.cpp:14-59."""
    _check(wavetype, mode, sphere)
    (h, a, b, r), t, single = _prep(thk, vp, vs, rho, period)
    nchain, n = h.shape
    ctx = default_context(device)
    c = np.zeros((nchain, len(t)))
    flag = np.zeros(nchain, dtype=np.int32)
    ctx.check(ctx.L.rfs_swd_forward(ctx.h, nchain, n, hptr(h), hptr(a), hptr(b), hptr(r), len(t), hptr(t),
                                    RFS_WAVE[wavetype], int(mode), int(bool(sphere)), hptr(c), hptr(flag)))
    if single:
        return c[0], bool(flag[0])
    return c, flag.astype(bool)


def adjoint_kernel(thk, vp, vs, rho, period, wavetype, mode=0, sphere=False, device=0):
    """(c, dcda, dcdb, dcdr, dcdh, bool) -- libsurf.adjoint_kernel, src/SWD/main.cpp:61-82."""
    _check(wavetype, mode, sphere)
    (h, a, b, r), t, single = _prep(thk, vp, vs, rho, period)
    nchain, n = h.shape
    ctx = default_context(device)
    nt = len(t)
    c = np.zeros((nchain, nt))
    ka, kb, kr, kh = (np.zeros((nchain, nt, n)) for _ in range(4))
    flag = np.zeros(nchain, dtype=np.int32)
    ctx.check(ctx.L.rfs_swd_kernel(ctx.h, nchain, n, hptr(h), hptr(a), hptr(b), hptr(r), nt, hptr(t),
                                   RFS_WAVE[wavetype], int(mode), int(bool(sphere)), hptr(c), hptr(ka), hptr(kb),
                                   hptr(kr), hptr(kh), hptr(flag)))
    if single:
        return c[0], ka[0], kb[0], kr[0], kh[0], bool(flag[0])
    return c, ka, kb, kr, kh, flag.astype(bool)
