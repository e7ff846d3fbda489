"""Drop-in for the reference's pybind11 extension ``librf`` (src/RF/main.cpp:191-213):
``forward``, ``kernel``, ``kernel_all`` with the reference's argument order (thk, rho, vp, vs,
qa, qb, ray_p, nt, dt, gauss, time_shift, method, water, rf_type[, par_type]), executed by the
HIP library.  2-D model arrays [nchain, nlayer] are accepted (outputs gain a chain axis).

Both methods: "time" (iterative time-domain deconvolution, the reference's default) and anything else =
"freq" (water-level spectral division), as src/RF/main.cpp:43,115,168 dispatch.

Deliberate differences: bad rf_type / par_type raise ValueError instead of exit(-1)
(main.cpp:40,101,162); traces of any length (those longer than 4096 samples take a
block-per-trace deconvolution kernel)."""
import numpy as np

from ..._lib import RfParams, default_context, hptr


def _params(ray_p, nt, dt, gauss, time_shift, method, water, rf_type, single_par=False):
    if rf_type in ("P", "p"):
        irf = 1
    elif rf_type in ("S", "s"):
        irf = 2
    else:
        raise ValueError("rf_type should be one of [P,p,S,s]")
    # RFS_RF_TIME = 0, RFS_RF_FREQ = 1, RFS_RF_TIME_PAR = 2 (frequency axis of cal_rf_par_time, RFModule.f90:27,47)
    imeth = (2 if single_par else 0) if method == "time" else 1
    return RfParams(float(ray_p), int(nt), float(dt), float(gauss), float(time_shift), float(water), irf, imeth)


def _prep(*arrs):
    out = [np.ascontiguousarray(np.asarray(a, dtype=np.float64)) for a in arrs]
    single = out[0].ndim == 1
    return [np.atleast_2d(a) for a in out], single


def forward(thk, rho, vp, vs, qa, qb, ray_p, nt, dt, gauss, time_shift, method="time", water=0.001,
            rf_type="P", device=0):
    """rf[nt] -- librf.forward, src/RF/main.cpp:17-62."""
    par = _params(ray_p, nt, dt, gauss, time_shift, method, water, rf_type)
    (h, r, a, b, qa_, qb_), single = _prep(thk, rho, vp, vs, qa, qb)
    nchain, n = h.shape
    ctx = default_context(device)
    rf = np.zeros((nchain, int(nt)))
    ctx.check(ctx.L.rfs_rf_forward(ctx.h, nchain, n, hptr(h), hptr(r), hptr(a), hptr(b), hptr(qa_), hptr(qb_),
                                   par, hptr(rf)))
    return rf[0] if single else rf


def kernel_all(thk, rho, vp, vs, qa, qb, ray_p, nt, dt, gauss, time_shift, method="time", water=0.001,
               rf_type="P", device=0, _single_par=False):
    """(rf[nt], k[4, nlayer, nt]), parameter axis [rho, vp, vs, thk] -- src/RF/main.cpp:140-189."""
    par = _params(ray_p, nt, dt, gauss, time_shift, method, water, rf_type, _single_par)
    (h, r, a, b, qa_, qb_), single = _prep(thk, rho, vp, vs, qa, qb)
    nchain, n = h.shape
    ctx = default_context(device)
    rf = np.zeros((nchain, int(nt)))
    kl = np.zeros((nchain, 4, n, int(nt)))
    ctx.check(ctx.L.rfs_rf_kernel_all(ctx.h, nchain, n, hptr(h), hptr(r), hptr(a), hptr(b), hptr(qa_), hptr(qb_),
                                      par, hptr(rf), hptr(kl)))
    return (rf[0], kl[0]) if single else (rf, kl)


_PAR = {"rho": 0, "vp": 1, "alpha": 1, "vs": 2, "beta": 2, "h": 3, "thick": 3}


def kernel(thk, rho, vp, vs, qa, qb, ray_p, nt, dt, gauss, time_shift, method="time", water=0.001,
           rf_type="P", par_type="vs", device=0):
    """(rf[nt], k[nlayer, nt]) for one parameter class -- src/RF/main.cpp:64-136."""
    if par_type not in _PAR:
        raise ValueError("par_type should be one of [vp,vs,rho,thick]")
    rf, kl = kernel_all(thk, rho, vp, vs, qa, qb, ray_p, nt, dt, gauss, time_shift, method, water, rf_type, device,
                        _single_par=True)
    return rf, np.ascontiguousarray(kl[..., _PAR[par_type], :, :])
