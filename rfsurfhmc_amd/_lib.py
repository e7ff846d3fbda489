"""ctypes binding of include/rfsurf.h (librfsurf_hip.so).  No fallback: a missing library or
a missing GPU raises."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.environ.get("RFSURF_LIB") or os.path.join(HERE, "librfsurf_hip.so")     # RFSURF_LIB: A/B builds of the same ABI

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int32_p = ctypes.POINTER(ctypes.c_int32)
c_float_p = ctypes.POINTER(ctypes.c_float)

RFS_WAVE = {"Rc": 0, "Rg": 1, "Lc": 2, "Lg": 3}
K_NAMES = ["prep", "rf_pass_a", "rf_mid", "rf_pass_b", "swd_roots", "swd_eigen", "combine", "swd_exact", "flow_step"]


class RfParams(ctypes.Structure):
    _fields_ = [("ray_p", ctypes.c_double), ("nt", ctypes.c_int32), ("dt", ctypes.c_double),
                ("gauss", ctypes.c_double), ("time_shift", ctypes.c_double), ("water", ctypes.c_double),
                ("rf_type", ctypes.c_int32), ("method", ctypes.c_int32)]


class SwdParams(ctypes.Structure):
    _fields_ = [("ntRc", ctypes.c_int32), ("ntRg", ctypes.c_int32), ("ntLc", ctypes.c_int32), ("ntLg", ctypes.c_int32),
                ("tRc", ctypes.c_void_p), ("tRg", ctypes.c_void_p), ("tLc", ctypes.c_void_p), ("tLg", ctypes.c_void_p),
                ("sphere", ctypes.c_int32), ("mode", ctypes.c_int32)]


class FlowNext(ctypes.Structure):      # rfs_flow_next (include/rfsurf.h): device pointers
    _fields_ = [(k, ctypes.c_void_p) for k in ("have", "u", "p", "rem", "xstart", "res_x", "res_val", "res_dsyn", "gsave",
                                               "kick")]


class FlowRecords(ctypes.Structure):   # rfs_flow_records
    _fields_ = [("buf", ctypes.c_void_p), ("bytes", ctypes.c_uint64), ("cap", ctypes.c_int32), ("want_dsyn", ctypes.c_int32),
                ("stamp", ctypes.c_double), ("reset", ctypes.c_int32)]


class RfsError(RuntimeError):
    pass


# every symbol include/rfsurf.h declares: (restype, argtypes)
_vp = ctypes.c_void_p
_i = ctypes.c_int
_d = ctypes.c_double
SIGNATURES = {
    "rfs_create": (_i, [ctypes.POINTER(_vp), _i, _i, _i]),
    "rfs_destroy": (None, [_vp]),
    "rfs_last_error": (ctypes.c_char_p, [_vp]),
    "rfs_set_stream": (_i, [_vp, _vp]),
    "rfs_synchronize": (_i, [_vp]),
    "rfs_swd_forward": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "rfs_swd_kernel": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rfs_rf_forward": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(RfParams), _vp]),
    "rfs_rf_kernel_all": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(RfParams), _vp, _vp]),
    "rfs_joint_setup": (_i, [_vp, _i, ctypes.POINTER(RfParams), _i, _vp, _i, _vp, _d, _d, _vp]),
    "rfs_joint_setup2": (_i, [_vp, _i, ctypes.POINTER(RfParams), ctypes.POINTER(SwdParams), _d, _d, _vp]),
    "rfs_joint_misfit_grad_dev": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "rfs_joint_misfit_grad": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "rfs_joint_forward": (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    "rfs_leapfrog_dev": (_i, [_vp, _i, _vp, _vp, _vp, _vp, ctypes.c_int32, _vp] + [_vp] * 8),
    "rfs_leapfrog_dev2": (_i, [_vp, _i, _vp, _vp, _vp, _vp, ctypes.c_int32, _vp, _vp] + [_vp] * 8),
    "rfs_flow_step": (_i, [_vp, _i] + [_vp] * 14),
    "rfs_flow_step2": (_i, [_vp, _i] + [_vp] * 14 + [ctypes.POINTER(FlowNext)]),
    "rfs_flow_step3": (_i, [_vp, _i] + [_vp] * 14 + [ctypes.POINTER(FlowNext), ctypes.POINTER(FlowRecords)]),
    "rfs_flow_deposit": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, ctypes.POINTER(FlowNext)]),
    "rfs_flow_restart": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp] + [_vp] * 7),
    "rfs_set_inverse_mass": (_i, [_vp, _vp]),
    "rfs_ndata": (_i, [_vp]),
    "rfs_set_option": (_i, [_vp, ctypes.c_char_p, _i]),
    "rfs_get_stat": (_i, [_vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]),
    "rfs_last_roots": (_i, [_vp, _i, c_int32_p, _vp]),
    "rfs_enable_timing": (_i, [_vp, _i]),
    "rfs_kernel_ms_sum": (_i, [_vp, _vp, _vp]),
    "rfs_kernel_timeline": (_i, [_vp, _vp, _vp, _vp]),
}
_DIAGNOSTIC = {"rfs_kernel_timeline", "rfs_last_roots", "rfs_flow_step3", "rfs_flow_deposit"}      # (absent from older builds loaded through RFSURF_LIB for A/B runs)

_LIB = None


def load() -> ctypes.CDLL:
    """Load librfsurf_hip.so and bind every declared symbol; raises if the library is absent."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIBPATH):
            raise RfsError(f"{LIBPATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)")
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and must be the one that gets loaded
        # (this library's DT_NEEDED libamdhip64.so.7 then resolves to it).  Loading /opt/rocm's copy first and
        # torch's afterwards leaves two runtimes in the process and torch then finds "No HIP GPUs".
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIBPATH)
        for name, (res, args) in SIGNATURES.items():
            if name in _DIAGNOSTIC and not hasattr(L, name) and os.environ.get("RFSURF_LIB"):
                continue
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def hptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


class Context:
    """Owns one rfs_ctx (device buffers, rocFFT plans, streams)."""

    def __init__(self, device: int = 0, max_chains: int = 8192, max_layers: int = 128):
        self.L = load()
        h = ctypes.c_void_p()
        rc = self.L.rfs_create(ctypes.byref(h), int(device), int(max_chains), int(max_layers))
        if rc != 0:
            raise RfsError(f"rfs_create failed ({rc}): no usable gfx950 device {device}? (no CPU fallback)")
        self.h = h
        self.device = device
        self.max_chains = max_chains

    def check(self, rc: int):
        if rc != 0:
            raise RfsError(f"librfsurf_hip error {rc}: {self.L.rfs_last_error(self.h).decode()}")

    def set_option(self, name: str, value: int):
        self.check(self.L.rfs_set_option(self.h, name.encode(), int(value)))

    def stat(self, name: str) -> int:
        v = ctypes.c_int64(0)
        self.check(self.L.rfs_get_stat(self.h, name.encode(), ctypes.byref(v)))
        return int(v.value)

    def last_roots(self, nchain: int) -> np.ndarray:
        """[nchain][items] roots of the last evaluation (rfs_last_roots: diagnostics)."""
        ni = ctypes.c_int32(0)
        self.check(self.L.rfs_last_roots(self.h, int(nchain), ctypes.byref(ni), None))
        out = np.zeros((int(nchain), int(ni.value)))
        self.check(self.L.rfs_last_roots(self.h, int(nchain), ctypes.byref(ni), hptr(out)))
        return out

    def close(self):
        if getattr(self, "h", None):
            self.L.rfs_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_DEFAULT = {}


def default_context(device: int = 0) -> Context:
    if device not in _DEFAULT:
        _DEFAULT[device] = Context(device=device, max_chains=1 << 20, max_layers=128)
    return _DEFAULT[device]
