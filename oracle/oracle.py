"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY.

Python face of the CPU oracle for the misfit+gradient hot path of nqdu/RfSurfHmc.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module; the product package ``rfsurfhmc_amd`` never does.

Three layers, each citing the reference file:line it restates:

* ``libsurf`` / ``librf`` -- objects with the call signatures of the reference's
  pybind11 extensions (src/SWD/main.cpp:14-93, src/RF/main.cpp:17-212), backed by
  the plain-C restatement in ``oracle/liboracle.so`` (swd_oracle.c, rf_oracle.c).
* ``SurfWD`` / ``ReceiverFunc`` / ``Joint_RF_SWD`` -- numpy restatement of the model
  plugins (model/model_surf.py:155-228, model/model_rf.py:137-198,
  model/model_rf_swd_vs_thk.py:66-86): empirical vp(vs), rho(vp), chain rule, K.r.
* ``ref_libsurf()`` / ``RefRFCore`` -- the reference itself, built from its own
  sources into ``oracle/_ref`` by ``oracle/Makefile`` (complete for src/SWD; the
  propagator/partials core only for src/RF, whose FFTW wrapper cannot be built
  here).  They validate the restatement and serve as CPU baseline.

Parity pin status: SWD -- pinned to the compiled reference (bit-exact phase
velocities, kernels to 1e-12) and to golden vectors generated from it.  RF --
every per-frequency quantity (R21, R22 and the 4*nlayer partials) pinned to the
compiled reference core; the 15-line water-level/IFFT tail is pinned only by the
known answers recorded in SURVEY.md section 8(c) and numpy's irfft.
"""
from __future__ import annotations

import ctypes
import importlib.util
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DP = ctypes.POINTER(ctypes.c_double)
_FP = ctypes.POINTER(ctypes.c_float)


def _d(a):
    return a.ctypes.data_as(_DP)


def _f(a):
    return a.ctypes.data_as(_FP)


def build(ref: bool = True) -> None:
    """Compile liboracle.so (always) and oracle/_ref (only where the reference is present)."""
    subprocess.run(["make", "-s", "-C", _HERE, "oracle"], check=True)
    if ref and os.path.isdir("/root/reference/src"):
        subprocess.run(["make", "-s", "-C", _HERE, "ref"], check=True)


_LIB = None


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = ctypes.CDLL(path)
        L.orc_nextpow2.restype = ctypes.c_int
        _LIB = L
    return _LIB


# --------------------------------------------------------------------------
# B1 level: same call signatures as the reference's pybind11 modules
# --------------------------------------------------------------------------
_WAVETYPES = {"Rc": 0, "Rg": 1, "Lc": 2, "Lg": 3}


class _LibSurf:
    """Restatement of libsurf (src/SWD/main.cpp:14-93): Rc/Rg/Lc/Lg, flat or spherical earth,
    fundamental mode.  Love kernels leave dcda at zero (the reference never writes it:
    surfdisp.cpp:258-296)."""

    nsec = 0  # secular-function evaluations of the last Rc/Rg flat call (work accounting)

    @staticmethod
    def _prep(thk, vp, vs, rho, period):
        f32 = [np.ascontiguousarray(np.asarray(a, dtype=np.float64).astype(np.float32))
               for a in (thk, vp, vs, rho)]  # forcecast to float32, main.cpp:9
        t = np.ascontiguousarray(np.asarray(period, dtype=np.float64))
        return f32, t

    def forward(self, thk, vp, vs, rho, period, wavetype, mode=0, sphere=False):
        if wavetype not in _WAVETYPES or mode < 0:
            raise NotImplementedError("oracle covers Rc/Rg/Lc/Lg")
        (h, a, b, r), t = self._prep(thk, vp, vs, rho, period)
        n, nt = len(h), len(t)
        cg = np.zeros(nt)
        L = lib()
        if mode > 0:           # higher modes: the mode loop of surfdisp96.f:227-316 (mode + 1 modes, the last one's roots remain)
            ierr = L.orc_swd_forward_m(_f(h), _f(a), _f(b), _f(r), n, _d(t), _d(cg), nt,
                                       _WAVETYPES[wavetype], int(bool(sphere)), int(mode))
        elif wavetype == "Rc" and not sphere:
            nsec = ctypes.c_long(0)
            ierr = L.orc_surfdisp_rc(_f(h), _f(a), _f(b), _f(r), n, _d(t), _d(cg), nt,
                                     ctypes.byref(nsec))
            _LibSurf.nsec = nsec.value
        else:
            ierr = L.orc_swd_forward(_f(h), _f(a), _f(b), _f(r), n, _d(t), _d(cg), nt,
                                     _WAVETYPES[wavetype], int(bool(sphere)))
        return cg, ierr != 1

    def adjoint_kernel(self, thk, vp, vs, rho, period, wavetype, mode=0, sphere=False):
        if wavetype not in _WAVETYPES or mode < 0:
            raise NotImplementedError("oracle covers Rc/Rg/Lc/Lg")
        (h, a, b, r), t = self._prep(thk, vp, vs, rho, period)
        n, nt = len(h), len(t)
        c = np.zeros(nt)
        ka, kb, kr, kh = (np.zeros((nt, n)) for _ in range(4))
        if mode > 0:
            ierr = lib().orc_swd_kernel_m(_f(h), _f(a), _f(b), _f(r), n, _d(t), _d(c), nt,
                                          _d(ka), _d(kb), _d(kr), _d(kh), _WAVETYPES[wavetype],
                                          int(bool(sphere)), int(mode))
        elif wavetype in ("Rc", "Rg") and not sphere:
            nsec = ctypes.c_long(0)
            ierr = lib().orc_surf_kernel(_f(h), _f(a), _f(b), _f(r), n, _d(t), _d(c), nt,
                                         _d(ka), _d(kb), _d(kr), _d(kh), _WAVETYPES[wavetype],
                                         ctypes.byref(nsec))
            _LibSurf.nsec = nsec.value
        else:
            ierr = lib().orc_swd_kernel(_f(h), _f(a), _f(b), _f(r), n, _d(t), _d(c), nt,
                                        _d(ka), _d(kb), _d(kr), _d(kh), _WAVETYPES[wavetype],
                                        int(bool(sphere)))
        return c, ka, kb, kr, kh, ierr != 1


def rayleigh_kernels_at_roots(thk, vp, vs, rho, period, c):
    """The reference's eigenfunction pass (sregn96, sregn96.f90:1637-1745, through orc_sregn96) at GIVEN phase velocities c
    instead of the ones its own root search would find: (dcda, dcdb, dcdr, dcdh) [nper][n] as _SurfKernel fills them for "Rc"
    (surfdisp.cpp:223-227).  Parity tests use it to separate what a root's value does to the kernels from everything else."""
    (h, a, b, r), t = _LibSurf._prep(thk, vp, vs, rho, period)
    n, nt = len(h), len(t)
    c = np.ascontiguousarray(np.asarray(c, dtype=np.float64))
    ka, kb, kr, kh = (np.zeros((nt, n)) for _ in range(4))
    L = lib()
    cg = ctypes.c_double(0.0)
    dummy = np.zeros(n)
    for k in range(nt):
        L.orc_sregn96(_f(h), _f(a), _f(b), _f(r), ctypes.c_int(n), ctypes.c_double(t[k]), ctypes.c_double(c[k]),
                      ctypes.byref(cg), _d(dummy), _d(dummy), _d(dummy), _d(dummy), _d(ka[k]), _d(kb[k]), _d(kh[k]), _d(kr[k]))
    return ka, kb, kr, kh


def swd_misfit_and_grad_at_roots(x, period, c, dobs):
    """SurfWD.misfit_and_grad (model/model_surf.py:155-228) for an Rc-only plugin with the roots GIVEN (see above)."""
    n = len(x) // 2
    vs, thk = x[:n], x[n:]
    vp, rho, dadb, drda = empirical_relation(vs)
    ka, kb, kr, kh = rayleigh_kernels_at_roots(thk, vp, vs, rho, period, c)
    kernel = kb + ka * dadb + kr * drda * dadb
    r = np.asarray(c, dtype=np.float64) - dobs
    return 0.5 * np.sum(r**2), np.hstack((r @ kernel, r @ kh))


def _rf_type(rf_type, time_shift):
    if rf_type in ("P", "p"):
        return 1, time_shift
    if rf_type in ("S", "s"):
        return 2, -time_shift  # main.cpp:35
    raise ValueError("rf_type should be one of [P,p,S,s]")


# --------------------------------------------------------------------------
# Time-domain receiver functions: iterative spike deconvolution (src/RF/deconit.f90) and its callers
# cal_rf_time / cal_rf_par_time(_all) (RFModule.f90:11-191).
# PARITY UNPINNED for this part: the reference computes every FFT through FFTW3 (src/RF/fftpack.f90), which
# this image lacks, so neither the reference's librf nor its deconit can be built here; what follows is a
# statement-by-statement numpy restatement (numpy.fft in place of the FFTW wrappers).  The per-frequency
# R21 / R22 / partials that feed it ARE pinned (compiled reference core, RefRFCore).
# --------------------------------------------------------------------------
PI32 = float(np.float32(np.arctan(np.float32(1.0))) * np.float32(4.0))     # atan(1.0) * 4. in default real


def _rfft(x, n):            # fftpack.f90:1-21
    return np.fft.rfft(x, n)


def _irfft(X, n):           # fftpack.f90:23-42 (c2r, then / n)
    return np.fft.irfft(X, n)


def gauss_filter(nt, dt, f0):
    """deconit.f90:15-32."""
    freq = np.arange(nt // 2 + 1) / (nt * dt)
    return np.exp(-0.25 * (2 * PI32 * freq / f0) ** 2)


def apply_gaussian(x, dt, f0):
    """deconit.f90:34-52."""
    n = len(x)
    return _irfft(_rfft(x, n) * gauss_filter(n, dt, f0), n)


def shift_data(x, dt, tshift):
    """deconit.f90:54-72."""
    n = len(x)
    i = np.arange(n // 2 + 1)
    return _irfft(_rfft(x, n) * np.exp(-1j * i / (n * dt) * PI32 * 2 * tshift), n)


def deconit(u, w, dt, tshift, f0, return_spikes=False):
    """deconit.f90:135-197: Ligorria & Ammon iterative time-domain deconvolution, at most 200 spikes."""
    nt = len(u)
    nft = 1
    while nft < nt:
        nft *= 2
    uflt = np.zeros(nft); wflt = np.zeros(nft)
    wflt[:nt] = w; uflt[:nt] = u
    wcopy = wflt.copy()
    uflt = apply_gaussian(uflt, dt, f0)
    wflt = apply_gaussian(wflt, dt, f0)
    with np.errstate(divide="ignore", invalid="ignore"):
        invpw = 1. / np.sum(wflt ** 2) / dt
        invpu = 1. / np.sum(uflt ** 2) / dt
    p = np.zeros(nft)
    sumsq_i = 1.0
    minderr = 0.001
    d_error = 100 * invpw + minderr
    rflt = uflt.copy()
    wf = _rfft(wflt, nft); wc = _rfft(wcopy, nft)
    spikes = []
    for _ in range(200):
        if abs(d_error) <= minderr:
            break
        cuw = _irfft(_rfft(rflt, nft) * np.conj(wf), nft) * dt            # mycorrelate
        idx = int(np.argmax(np.abs(cuw[:nft // 2])))                      # maxloc: first maximum
        p[idx] = p[idx] + cuw[idx] * invpw / dt
        spikes.append(idx)
        temp1 = apply_gaussian(p.copy(), dt, f0)
        temp2 = _irfft(_rfft(temp1, nft) * wc, nft)                       # myconvolve
        rflt = uflt - temp2 * dt
        with np.errstate(invalid="ignore"):
            sumsq = np.sum(rflt ** 2) * dt * invpu
        d_error = 100. * (sumsq_i - sumsq)
        sumsq_i = sumsq
    out = shift_data(apply_gaussian(p, dt, f0), dt, tshift)[:nt]
    return (out, spikes) if return_spikes else out


def rf_time_from_spectra(R21, R22, nft, nt, dt, f0, tshift):
    """tail of cal_rf_time, RFModule.f90:180-186."""
    ux = _irfft(R22, nft); uz = _irfft(R21, nft)
    return deconit(ux, uz, dt, tshift, f0)[:nt]


def rf_par_time_from_spectra(R21, R22, R21m, R22m, nft, nt, dt, f0, tshift):
    """tail of cal_rf_par_time_all, RFModule.f90:120-137.  R21m/R22m: [n2, 4, nlayer]."""
    rf = rf_time_from_spectra(R21, R22, nft, nt, dt, f0, tshift)
    uz = _irfft(R21 ** 2, nft)
    n = R21m.shape[2]
    kl = np.zeros((4, n, nt))
    for ip in range(4):
        for j in range(n):
            num = R22m[:, ip, j] * R21 - R21m[:, ip, j] * R22
            kl[ip, j] = deconit(_irfft(num, nft), uz, dt, tshift, f0)[:nt]
    return rf, kl


def _time_axis(nft, dt, n2, which):
    """omega(it): cal_rf_time :173 and cal_rf_par_time :47 use atan(1.0)*4.0 (float32 pi),
    cal_rf_par_time_all :112 uses atan(1.0_dp)*4.0_dp."""
    it = np.arange(n2)
    if which == "forward":
        return (1.0 / dt / nft) * it * 2 * PI32
    if which == "kernel":
        return 1.0 / nft / dt * it * 2.0 * PI32
    return 1.0 / nft / dt * it * 2.0 * (np.arctan(1.0) * 4.0)


class _LibRF:
    """Restatement of librf (src/RF/main.cpp:17-212): frequency-domain method in C (rf_oracle.c); time-domain
    method = C restatement of the propagator stack + the numpy restatement of deconit above."""

    def _spectra_time(self, thk, rho, vp, vs, qa, qb, ray_p, nt, dt, irf, which, partials):
        n = len(thk)
        nft = lib().orc_nextpow2(int(nt))
        n2 = nft // 2 + 1
        al = np.ascontiguousarray(vp * (1.0 + 1j / (2.0 * qa) + 1.0 / (8.0 * qa**2)))
        be = np.ascontiguousarray(vs * (1.0 + 1j / (2.0 * qb) + 1.0 / (8.0 * qb**2)))
        w = _time_axis(nft, dt, n2, which)
        R21 = np.zeros(n2, complex); R22 = np.zeros(n2, complex)
        R21m = np.zeros((n2, 4, n), complex); R22m = np.zeros((n2, 4, n), complex)
        c = ctypes.c_double
        for it in range(n2):
            if partials:
                lib().orcprobe_rf_response_par_all(c(w[it]), c(0.0), c(ray_p), n, _d(thk), _d(al), _d(be), _d(vp),
                                                   _d(vs), _d(rho), irf, _d(R21[it:]), _d(R22[it:]), _d(R21m[it]),
                                                   _d(R22m[it]))
            else:
                lib().orcprobe_rf_response(c(w[it]), c(0.0), c(ray_p), n, _d(thk), _d(al), _d(be), _d(rho), irf,
                                           _d(R21[it:]), _d(R22[it:]))
        return nft, R21, R22, R21m, R22m

    @staticmethod
    def _prep(*arrs):
        return [np.ascontiguousarray(np.asarray(a, dtype=np.float64)) for a in arrs]

    def forward(self, thk, rho, vp, vs, qa, qb, ray_p, nt, dt, gauss, time_shift,
                method="time", water=0.001, rf_type="P"):
        irf, t0 = _rf_type(rf_type, time_shift)
        thk, rho, vp, vs, qa, qb = self._prep(thk, rho, vp, vs, qa, qb)
        if method == "time":
            nft, R21, R22, _, _ = self._spectra_time(thk, rho, vp, vs, qa, qb, ray_p, nt, dt, irf, "forward", False)
            return rf_time_from_spectra(R21, R22, nft, int(nt), dt, gauss, t0)
        rf = np.zeros(nt)
        c = ctypes.c_double
        lib().orc_rf_freq(_d(thk), _d(vp), _d(vs), _d(rho), _d(qa), _d(qb), len(thk), int(nt),
                          c(dt), c(ray_p), c(gauss), c(t0), c(water), irf, _d(rf))
        return rf

    def kernel_all(self, thk, rho, vp, vs, qa, qb, ray_p, nt, dt, gauss, time_shift,
                   method="time", water=0.001, rf_type="P"):
        irf, t0 = _rf_type(rf_type, time_shift)
        thk, rho, vp, vs, qa, qb = self._prep(thk, rho, vp, vs, qa, qb)
        if method == "time":
            nft, R21, R22, R21m, R22m = self._spectra_time(thk, rho, vp, vs, qa, qb, ray_p, nt, dt, irf,
                                                           getattr(self, "_axis", "kernel_all"), True)
            return rf_par_time_from_spectra(R21, R22, R21m, R22m, nft, int(nt), dt, gauss, t0)
        n = len(thk)
        rf = np.zeros(nt)
        kl = np.zeros((4, n, nt))
        c = ctypes.c_double
        lib().orc_rf_par_freq_all(_d(thk), _d(vp), _d(vs), _d(rho), _d(qa), _d(qb), n, int(nt),
                                  c(dt), c(ray_p), c(gauss), c(t0), c(water), irf, _d(rf), _d(kl))
        return rf, kl

    def kernel(self, thk, rho, vp, vs, qa, qb, ray_p, nt, dt, gauss, time_shift,
               method="time", water=0.001, rf_type="P", par_type="vs"):
        idx = {"rho": 0, "vp": 1, "alpha": 1, "vs": 2, "beta": 2, "h": 3, "thick": 3}[par_type]
        self._axis = "kernel"          # cal_rf_par_time's frequency axis uses the float32 pi (RFModule.f90:27,47)
        try:
            rf, kl = self.kernel_all(thk, rho, vp, vs, qa, qb, ray_p, nt, dt, gauss, time_shift,
                                     method, water, rf_type)
        finally:
            self._axis = "kernel_all"
        return rf, kl[idx].copy()


libsurf = _LibSurf()
librf = _LibRF()


# --------------------------------------------------------------------------
# B2 level: numpy restatement of the model plugins
# --------------------------------------------------------------------------
def empirical_relation(vs):
    """model/model_surf.py:47-79 == model/model_rf.py:52-77: (vp, rho, dadb, drda)."""
    vp = 0.9409 + 2.0947 * vs - 0.8206 * vs**2 + 0.2683 * vs**3 - 0.0251 * vs**4
    rho = 1.6612 * vp - 0.4721 * vp**2 + 0.0671 * vp**3 - 0.0043 * vp**4 + 0.000106 * vp**5
    drda = 1.6612 - 0.4721 * 2 * vp + 0.0671 * 3 * vp**2 - 0.0043 * 4 * vp**3 + 0.000106 * 5 * vp**4
    dadb = 2.0947 - 0.8206 * 2 * vs + 0.2683 * 3 * vs**2 - 0.0251 * 4 * vs**3
    return vp, rho, dadb, drda


class SurfWD:
    """model/model_surf.py.  ``lib`` = any object with the libsurf API.  Keeps the reference's period quirks:
    forward() computes every block at tRc (:104-131); misfit_and_grad() computes Lc and Lg at tRc (:199-216)."""

    def __init__(self, tRc=None, tRg=None, lib=None, mode=0, sphere=False, tLc=None, tLg=None):
        self.lib = lib if lib is not None else libsurf
        self.mode, self.sphere = mode, sphere
        for name, t in (("tRc", tRc), ("tRg", tRg), ("tLc", tLc), ("tLg", tLg)):
            arr = np.asarray(t, dtype=float) if t is not None and len(t) > 0 else None
            setattr(self, name, arr)
            setattr(self, "n" + name, 0 if arr is None else len(arr))
        self.nt = self.ntRc + self.ntRg + self.ntLc + self.ntLg

    def set_obsdata(self, dobs):
        self.dobs = dobs

    def forward(self, x):
        """model_surf.py:81-133 -- note every block is computed at tRc (quirk :104-131)."""
        n = len(x) // 2
        vs, thk = x[:n], x[n:]
        vp, rho, _, _ = empirical_relation(vs)
        d = np.zeros(self.nt)
        k1 = 0
        for nrow, wt in ((self.ntRc, "Rc"), (self.ntRg, "Rg"), (self.ntLc, "Lc"), (self.ntLg, "Lg")):
            if nrow == 0:
                continue
            d[k1:k1 + nrow], flag = self.lib.forward(thk, vp, vs, rho, self.tRc, wt, self.mode, self.sphere)
            if not flag:
                return d, flag
            k1 += nrow
        return d, True

    def misfit_and_grad(self, x):
        """model_surf.py:155-228.  Love kernels: dcda is uninitialised memory in the reference
        (surfdisp.cpp:258-296); it is taken as zero here."""
        n = len(x) // 2
        vs, thk = x[:n], x[n:]
        vp, rho, dadb, drda = empirical_relation(vs)
        kernel = np.zeros((self.nt, n))
        kernel_thk = np.zeros((self.nt, n))
        d = np.zeros(self.nt)
        k1 = 0
        for nrow, periods, wt in ((self.ntRc, self.tRc, "Rc"), (self.ntRg, self.tRg, "Rg"),
                                  (self.ntLc, self.tRc, "Lc"), (self.ntLg, self.tRc, "Lg")):
            if nrow == 0:
                continue
            k2 = k1 + nrow
            cg, dcda, dcdb, dcdr, dcdh, flag = self.lib.adjoint_kernel(
                thk, vp, vs, rho, periods, wt, self.mode, self.sphere)
            if not flag:
                return 0.0, np.zeros(n), np.zeros(self.nt), False
            if wt[0] == "L":
                dcda = np.zeros_like(dcdb)
            d[k1:k2] = cg
            kernel[k1:k2] = dcdb + dcda * dadb + dcdr * drda * dadb
            kernel_thk[k1:k2] = dcdh
            k1 = k2
        r = d - self.dobs
        grad = np.hstack((r @ kernel, r @ kernel_thk))
        return 0.5 * np.sum(r**2), grad, d, True


class ReceiverFunc:
    """model/model_rf.py.  ``lib`` = any object with the librf API."""

    def __init__(self, ray_p, nt, dt, gauss, time_shift, water_level=0.001, type_="P",
                 method="freq", lib=None):
        self.lib = lib if lib is not None else librf
        self.ray_p, self.nt, self.dt, self.gauss = ray_p, nt, dt, gauss
        self.time_shift, self.water_level, self.rf_type, self.method = time_shift, water_level, type_, method

    def set_obsdata(self, dobs):
        self.dobs = dobs

    def forward(self, x):
        """model_rf.py:79-116."""
        n = len(x) // 2
        vs, thk = x[:n], x[n:]
        vp, rho, _, _ = empirical_relation(vs)
        qa = thk * 0 + 9999.0
        return self.lib.forward(thk, rho, vp, vs, qa, qa.copy(), self.ray_p, self.nt, self.dt,
                                self.gauss, self.time_shift, self.method, self.water_level, self.rf_type)

    def misfit_and_grad(self, x):
        """model_rf.py:137-198 (3-tuple, no flag)."""
        n = len(x) // 2
        vs, thk = x[:n], x[n:]
        vp, rho, dadb, drda = empirical_relation(vs)
        qa = thk * 0 + 9999.0
        d, kl = self.lib.kernel_all(thk, rho, vp, vs, qa, qa.copy(), self.ray_p, self.nt, self.dt,
                                    self.gauss, self.time_shift, self.method, self.water_level,
                                    self.rf_type)
        krho, kvp, kvs, kthk = kl[0], kl[1], kl[2], kl[3]
        kernel = kvs + dadb[:, None] * kvp + (drda * dadb)[:, None] * krho
        r = d - self.dobs
        grad = np.hstack((kernel @ r, kthk @ r))
        return 0.5 * np.sum(r**2), grad, d


class Joint_RF_SWD:
    """model/model_rf_swd_vs_thk.py."""

    def __init__(self, sigma1, sigma2, rfmodel, swdmodel):
        self.sigma1, self.sigma2, self.rfmodel, self.swdmodel = sigma1, sigma2, rfmodel, swdmodel
        self.ndata = rfmodel.nt + swdmodel.nt

    def set_obsdata(self, rfobs, swdobs):
        self.rfmodel.set_obsdata(np.asarray(rfobs) * 1.0)
        self.swdmodel.set_obsdata(np.asarray(swdobs) * 1.0)
        self.dobs = np.concatenate((self.rfmodel.dobs, self.swdmodel.dobs))

    def forward(self, x):
        drf = self.rfmodel.forward(x)
        dswd, flag = self.swdmodel.forward(x)
        return drf, dswd, flag

    def misfit_and_grad(self, x):
        """model_rf_swd_vs_thk.py:66-86."""
        misfitr, gradr, dr = self.rfmodel.misfit_and_grad(x)
        misfits, grads, ds, flag = self.swdmodel.misfit_and_grad(x)
        if not flag:
            return 0.0, np.zeros(gradr.shape), self.dobs, flag
        wt = (self.sigma1 / self.sigma2) ** 2 * dr.size / ds.size
        return misfitr + wt * misfits, gradr + wt * grads, np.concatenate((dr, ds)), True


# --------------------------------------------------------------------------
# the reference itself (oracle/_ref), when it has been built
# --------------------------------------------------------------------------
def ref_available() -> bool:
    d = os.path.join(_HERE, "_ref")
    return os.path.isdir(d) and any(f.startswith("libsurf") for f in os.listdir(d)) \
        and os.path.exists(os.path.join(d, "librf_core_ref.so"))


def ref_librf_full():
    """The reference's COMPLETE pybind11 module ``librf`` (oracle/Makefile builds it only where FFTW3 exists;
    not in this image) or None.  With it the RF tail -- water level, Gaussian, irfft, e^{sigma t} scaling
    (RFModule.f90:392-425) -- and ``deconit`` are pinned by reference runs instead of restated."""
    d = os.path.join(_HERE, "_ref")
    if not os.path.isdir(d):
        return None
    names = [f for f in os.listdir(d) if f.startswith("librf.") and f.endswith(".so")]
    if not names:
        return None
    spec = importlib.util.spec_from_file_location("librf", os.path.join(d, names[0]))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ref_libsurf():
    """The reference's own pybind11 module ``libsurf`` (complete build of src/SWD)."""
    import sysconfig
    suffix = sysconfig.get_config_var("EXT_SUFFIX")         # _ref/ holds this interpreter's build; _ref/py39/ the
    d = os.path.join(_HERE, "_ref")                         # one for the conda python3.9 that has a real h5py
    hits = [os.path.join(r, f) for r in (d, os.path.join(d, "py39")) if os.path.isdir(r)
            for f in os.listdir(r) if f == "libsurf" + suffix]
    if not hits:
        raise ImportError(f"oracle/_ref holds no libsurf{suffix} (make -C oracle ref)")
    spec = importlib.util.spec_from_file_location("libsurf", hits[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ref_libsurf_variant(kind: str):
    """The reference's libsurf at another optimisation setting (oracle/Makefile `ref_variants`): "O2" = the pin above,
    "O0", "native" (-O3 -march=native: the reference's own Release flags, CMakeLists.txt:16-34).  Not an oracle -- used by
    oracle/make_golden.py --ill-conditioned and scripts/ref_selfdiff.py to measure the reference against itself.
    ONE variant per process: the interpreter keeps one extension module per name (a second `libsurf` comes back as the
    first), so callers that compare builds run each in a process of its own."""
    import sysconfig
    if kind == "O2":
        return ref_libsurf()
    path = os.path.join(_HERE, "_ref", kind, "libsurf" + sysconfig.get_config_var("EXT_SUFFIX"))
    if not os.path.exists(path):
        raise ImportError(f"{path} not built (make -C oracle ref_variants)")
    if "libsurf_ref_loaded" in globals() and globals()["libsurf_ref_loaded"] != path:
        raise RuntimeError("one build of the reference's libsurf per process")
    globals()["libsurf_ref_loaded"] = path
    spec = importlib.util.spec_from_file_location("libsurf", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert os.path.samefile(mod.__file__, path), mod.__file__
    return mod


class RefRFCore:
    """Compiled reference RF core (RFModule.f90 procedures) behind oracle/ref_probe.c.

    ``kernel_all``/``forward`` combine the reference's per-frequency R21/R22/partials
    with this file's restatement of the reference's 15-line tail (RFModule.f90:392-425)
    using numpy's irfft -- a HYBRID, labelled as such wherever its output is stored.
    """

    def __init__(self):
        self.L = ctypes.CDLL(os.path.join(_HERE, "_ref", "librf_core_ref.so"))
        self.L.refprobe_nextpow2.restype = ctypes.c_int

    @staticmethod
    def _atten(v, q):
        return np.ascontiguousarray(v * (1.0 + 1j / (2.0 * q) + 1.0 / (8.0 * q**2)))

    def spectra_time(self, thk, rho, vp, vs, qa, qb, ray_p, nt, dt, rf_type, which, partials):
        """R21, R22 (and partials) on the real frequency axis of the time-domain method, from the compiled
        reference propagator core."""
        thk, rho, vp, vs, qa, qb = (np.ascontiguousarray(np.asarray(a, dtype=float))
                                    for a in (thk, rho, vp, vs, qa, qb))
        n = len(thk)
        nft = self.L.refprobe_nextpow2(int(nt))
        n2 = nft // 2 + 1
        al, be = self._atten(vp, qa), self._atten(vs, qb)
        w = _time_axis(nft, dt, n2, which)
        R21 = np.zeros(n2, complex); R22 = np.zeros(n2, complex)
        R21m = np.zeros((n2, 4, n), complex); R22m = np.zeros((n2, 4, n), complex)
        c = ctypes.c_double
        for it in range(n2):
            if partials:
                self.L.refprobe_rf_response_par_all(
                    c(w[it]), c(0.0), c(ray_p), n, _d(thk), _d(al), _d(be), _d(vp), _d(vs),
                    _d(rho), rf_type, _d(R21[it:]), _d(R22[it:]), _d(R21m[it]), _d(R22m[it]))
            else:
                self.L.refprobe_rf_response(
                    c(w[it]), c(0.0), c(ray_p), n, _d(thk), _d(al), _d(be), _d(rho), rf_type,
                    _d(R21[it:]), _d(R22[it:]))
        return nft, R21, R22, R21m, R22m

    def spectra(self, thk, rho, vp, vs, qa, qb, ray_p, nt, dt, rf_type=1, partials=True):
        thk, rho, vp, vs, qa, qb = (np.ascontiguousarray(np.asarray(a, dtype=float))
                                    for a in (thk, rho, vp, vs, qa, qb))
        n = len(thk)
        nft = self.L.refprobe_nextpow2(int(nt))
        n2 = nft // 2 + 1
        al, be = self._atten(vp, qa), self._atten(vs, qb)
        pi32 = float(np.float32(np.arctan(np.float32(1.0)) * np.float32(4.0)))
        sigma = 1.0 / dt / nft * 4.0
        w = np.array([1.0 / nft / dt * it * 2.0 * pi32 for it in range(n2)])
        R21 = np.zeros(n2, complex)
        R22 = np.zeros(n2, complex)
        R21m = np.zeros((n2, 4, n), complex)
        R22m = np.zeros((n2, 4, n), complex)
        c = ctypes.c_double
        for it in range(n2):
            if partials:
                self.L.refprobe_rf_response_par_all(
                    c(w[it]), c(-sigma), c(ray_p), n, _d(thk), _d(al), _d(be), _d(vp), _d(vs),
                    _d(rho), rf_type, _d(R21[it:]), _d(R22[it:]), _d(R21m[it]), _d(R22m[it]))
            else:
                self.L.refprobe_rf_response(
                    c(w[it]), c(-sigma), c(ray_p), n, _d(thk), _d(al), _d(be), _d(rho), rf_type,
                    _d(R21[it:]), _d(R22[it:]))
        return w, sigma, nft, R21, R22, R21m, R22m

    @staticmethod
    def _tail(spec, nft, nt, dt, sigma, t0):
        tr = np.fft.irfft(spec, nft, axis=-1)[..., :nt]
        return tr / dt * np.exp(sigma * (-t0 + np.arange(nt) * dt))

    def kernel_all(self, thk, rho, vp, vs, qa, qb, ray_p, nt, dt, gauss, time_shift,
                   method="freq", water=0.001, rf_type="P"):
        irf, t0 = _rf_type(rf_type, time_shift)
        if method == "time":
            nft, R21, R22, R21m, R22m = self.spectra_time(thk, rho, vp, vs, qa, qb, ray_p, nt, dt, irf, "kernel_all", True)
            return rf_par_time_from_spectra(R21, R22, R21m, R22m, nft, int(nt), dt, gauss, t0)
        w, sigma, nft, R21, R22, R21m, R22m = self.spectra(thk, rho, vp, vs, qa, qb, ray_p, nt, dt, irf)
        g = np.exp(-(w / 2 / gauss) ** 2) * np.exp(-1j * w * t0)
        wa = (R21 * np.conj(R21)).real
        fai = np.maximum(wa, water * wa.max())
        rf = self._tail(np.conj(R21) * R22 * g / fai, nft, nt, dt, sigma, t0)
        sq = R21**2
        wa = (sq * np.conj(sq)).real
        fai = np.maximum(wa, water * wa.max())
        spec = (np.conj(sq) * g / fai)[:, None, None] * (R22m * R21[:, None, None] - R21m * R22[:, None, None])
        kl = self._tail(np.moveaxis(spec, 0, -1), nft, nt, dt, sigma, t0)
        return rf, np.ascontiguousarray(kl)

    def forward(self, thk, rho, vp, vs, qa, qb, ray_p, nt, dt, gauss, time_shift,
                method="freq", water=0.001, rf_type="P"):
        irf, t0 = _rf_type(rf_type, time_shift)
        if method == "time":
            nft, R21, R22, _, _ = self.spectra_time(thk, rho, vp, vs, qa, qb, ray_p, nt, dt, irf, "forward", False)
            return rf_time_from_spectra(R21, R22, nft, int(nt), dt, gauss, t0)
        w, sigma, nft, R21, R22, _, _ = self.spectra(thk, rho, vp, vs, qa, qb, ray_p, nt, dt, irf,
                                                     partials=False)
        g = np.exp(-(w / 2 / gauss) ** 2) * np.exp(-1j * w * t0)
        wa = (R21 * np.conj(R21)).real
        fai = np.maximum(wa, water * wa.max())
        return self._tail(np.conj(R21) * R22 * g / fai, nft, nt, dt, sigma, t0)
