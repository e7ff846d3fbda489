"""HDF5 files without h5py: the handful of libhdf5 C calls the result store needs, bound with ctypes.

The reference writes its chains through h5py (pyhmc/hmc.py:58 ``h5py.File(..., "w")``, :203-226 ``create_group`` /
``create_dataset`` / ``fio[name][:] = x``) and ``src/plot_results.py:106-156`` reads them back by member name.  This
module offers the same small surface -- ``File(path, mode)``, ``create_group``, ``create_dataset(name, data= | shape=,
dtype=)``, ``f[name][:]``, ``f[name][:] = x``, ``keys()``, ``in`` -- over the HDF5 C library itself, so that the
files are genuine HDF5 whatever Python the sampler runs under.  ``open_file`` hands out ``h5py.File`` where h5py is
importable and this binding otherwise; without either it raises ImportError naming what it looked for.

Library search: ``$RFSURF_HDF5_LIB`` (a path), ``ctypes.util.find_library("hdf5")``, then the usual lib directories
of the running prefix, the system and a conda installation.  HDF5 1.8 (32-bit ``hid_t``) and 1.10+ (64-bit) are both
handled; only native little-endian f64 / f32 / i64 / i32 / u8 datasets of fixed shape are supported -- all the
store ever writes."""
from __future__ import annotations

import ctypes
import ctypes.util
import glob
import os
import sys

import numpy as np

_H5F_ACC_RDONLY, _H5F_ACC_RDWR, _H5F_ACC_TRUNC = 0, 1, 2
_H5S_SELECT_SET = 0
_H5T_INTEGER, _H5T_FLOAT = 0, 1
_H5I_GROUP, _H5I_DATASET = 2, 5
_H5_INDEX_NAME, _H5_ITER_INC = 0, 0

_lib = None


class _GInfo(ctypes.Structure):          # H5G_info_t
    _fields_ = [("storage_type", ctypes.c_int), ("nlinks", ctypes.c_uint64), ("max_corder", ctypes.c_int64),
                ("mounted", ctypes.c_int)]


def _candidates():
    env = os.environ.get("RFSURF_HDF5_LIB")
    if env:
        yield env
        return
    found = ctypes.util.find_library("hdf5") or ctypes.util.find_library("hdf5_serial")
    if found:
        yield found
    pats = [os.path.join(sys.prefix, "lib", "libhdf5.so*"), "/usr/lib/x86_64-linux-gnu/libhdf5.so*",
            "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*",
            "/usr/lib64/libhdf5.so*", "/usr/local/lib/libhdf5.so*", "/opt/conda/lib/libhdf5.so*"]
    for p in pats:
        for f in sorted(glob.glob(p)):
            yield f


class _Lib:
    """libhdf5 with argument types set for the ``hid_t`` width of the version that was found."""

    def __init__(self, path):
        L = self.L = ctypes.CDLL(path)
        self.path = path
        maj, mnr, rel = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
        if L.H5open() < 0 or L.H5get_libversion(ctypes.byref(maj), ctypes.byref(mnr), ctypes.byref(rel)) < 0:
            raise OSError(f"{path}: H5open failed")
        self.version = (maj.value, mnr.value, rel.value)
        hid = self.hid = ctypes.c_int64 if self.version >= (1, 10, 0) else ctypes.c_int32
        cp, ci, vp = ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p
        u64p = ctypes.POINTER(ctypes.c_uint64)
        sig = {
            "H5Fcreate": (hid, [cp, ctypes.c_uint, hid, hid]), "H5Fopen": (hid, [cp, ctypes.c_uint, hid]),
            "H5Fclose": (ci, [hid]), "H5Fflush": (ci, [hid, ci]),
            "H5Gcreate2": (hid, [hid, cp, hid, hid, hid]), "H5Gopen2": (hid, [hid, cp, hid]), "H5Gclose": (ci, [hid]),
            "H5Gget_info": (ci, [hid, ctypes.POINTER(_GInfo)]),
            "H5Screate_simple": (hid, [ci, u64p, u64p]), "H5Screate": (hid, [ci]), "H5Sclose": (ci, [hid]),
            "H5Sget_simple_extent_ndims": (ci, [hid]), "H5Sget_simple_extent_dims": (ci, [hid, u64p, u64p]),
            "H5Sselect_hyperslab": (ci, [hid, ci, u64p, u64p, u64p, u64p]),
            "H5Dcreate2": (hid, [hid, cp, hid, hid, hid, hid, hid]), "H5Dopen2": (hid, [hid, cp, hid]),
            "H5Dclose": (ci, [hid]), "H5Dget_space": (hid, [hid]), "H5Dget_type": (hid, [hid]),
            "H5Dwrite": (ci, [hid, hid, hid, hid, hid, vp]), "H5Dread": (ci, [hid, hid, hid, hid, hid, vp]),
            "H5Tget_class": (ci, [hid]), "H5Tget_size": (ctypes.c_size_t, [hid]), "H5Tget_sign": (ci, [hid]),
            "H5Tclose": (ci, [hid]),
            "H5Lexists": (ci, [hid, cp, hid]),
            "H5Lget_name_by_idx": (ctypes.c_ssize_t, [hid, cp, ci, ci, ctypes.c_uint64, cp, ctypes.c_size_t, hid]),
            "H5Oopen": (hid, [hid, cp, hid]), "H5Oclose": (ci, [hid]), "H5Iget_type": (ci, [hid]),
            "H5Eset_auto2": (ci, [hid, vp, vp]),
            "H5Pcreate": (hid, [hid]), "H5Pset_fclose_degree": (ci, [hid, ci]), "H5Pclose": (ci, [hid]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        self.types = {np.dtype(k): hid.in_dll(L, v).value for k, v in (
            ("f8", "H5T_NATIVE_DOUBLE_g"), ("f4", "H5T_NATIVE_FLOAT_g"), ("i8", "H5T_NATIVE_INT64_g"),
            ("i4", "H5T_NATIVE_INT32_g"), ("u1", "H5T_NATIVE_UINT8_g"))}
        L.H5Eset_auto2(0, None, None)    # errors come back as return codes and are raised here, not printed
        # file-access class id: H5P_CLS_FILE_ACCESS_ID_g (>= 1.8.15 / 1.10), H5P_CLS_FILE_ACCESS_g before
        self.fapl_class = None
        for sym in ("H5P_CLS_FILE_ACCESS_ID_g", "H5P_CLS_FILE_ACCESS_g"):
            try:
                self.fapl_class = hid.in_dll(L, sym).value
                break
            except ValueError:
                continue

    def strong_close_fapl(self):
        """File-access property list with H5F_CLOSE_STRONG: closing the file closes every dataset / group handle still
        open in it, as h5py's File.close() does -- with the library's default (weak) degree a caller that still holds a
        Dataset would keep the file open and a later File(path, "w") would fail.  0 (default list) where unavailable."""
        if self.fapl_class is None:
            return 0
        fapl = self.L.H5Pcreate(self.fapl_class)
        if fapl < 0:
            return 0
        if self.L.H5Pset_fclose_degree(fapl, 3) < 0:          # H5F_CLOSE_STRONG
            self.L.H5Pclose(fapl)
            return 0
        return fapl


def library():
    """The loaded libhdf5 wrapper, or ImportError listing where it was looked for."""
    global _lib
    if _lib is None:
        tried = []
        for path in _candidates():
            try:
                _lib = _Lib(path)
                break
            except (OSError, AttributeError, ValueError) as e:
                tried.append(f"{path} ({e})")
        else:
            raise ImportError("no usable HDF5: h5py is not importable and libhdf5 was not found "
                              "(set RFSURF_HDF5_LIB to the library's path); tried: " + ("; ".join(tried) or "nothing on the search path"))
    return _lib


def backend():
    """"h5py", "libhdf5" or None -- what ``open_file`` would use."""
    try:
        import h5py  # noqa: F401
        return "h5py"
    except ImportError:
        pass
    try:
        library()
        return "libhdf5"
    except ImportError:
        return None


def _check(rc, what):
    if rc < 0:
        raise OSError(f"HDF5: {what} failed")
    return rc


def _dims(seq):
    return (ctypes.c_uint64 * len(seq))(*seq) if len(seq) else None


class Dataset:
    """Fixed-shape dataset: ``d[:]`` / ``d[...]`` / ``d[i]`` read one hyperslab (any other index reads everything and
    lets numpy slice), ``d[:] = a`` / ``d[...] = a`` / ``d[i] = a`` write."""
    _id = None

    def __init__(self, lib, did, name):
        self._lib, self._id, self.name = lib, did, name
        L = lib.L
        sp = _check(L.H5Dget_space(did), "H5Dget_space")
        nd = _check(L.H5Sget_simple_extent_ndims(sp), "H5Sget_simple_extent_ndims")
        dims = (ctypes.c_uint64 * max(nd, 1))()
        if nd:
            _check(L.H5Sget_simple_extent_dims(sp, dims, None), "H5Sget_simple_extent_dims")
        L.H5Sclose(sp)
        self.shape = tuple(int(dims[i]) for i in range(nd))
        tp = _check(L.H5Dget_type(did), "H5Dget_type")
        cls, size, sign = L.H5Tget_class(tp), L.H5Tget_size(tp), L.H5Tget_sign(tp)
        L.H5Tclose(tp)
        if cls == _H5T_FLOAT and size in (4, 8):
            self.dtype = np.dtype(f"f{size}")
        elif cls == _H5T_INTEGER and size in (1, 4, 8):
            self.dtype = np.dtype(("i" if sign else "u") + str(size)) if size > 1 else np.dtype("u1")
        else:
            raise TypeError(f"{name}: HDF5 type class {cls} of {size} bytes is outside what this binding reads")
        if self.dtype not in lib.types:
            raise TypeError(f"{name}: no native type for {self.dtype}")

    def close(self):
        if self._id is not None:
            self._lib.L.H5Dclose(self._id)
            self._id = None

    def __del__(self):
        try:
            self.close()
        except Exception:            # interpreter shutdown: the library may already be gone
            pass

    def _row(self, key):
        """(file space, mem space, shape) of row ``key`` of the first axis, or (0, 0, shape) for everything."""
        if isinstance(key, (int, np.integer)):
            pass
        elif key is Ellipsis or (isinstance(key, slice) and key == slice(None)) or (isinstance(key, tuple) and len(key) == 0):
            return 0, 0, self.shape
        if not isinstance(key, (int, np.integer)) or not self.shape:
            raise IndexError("only [:], [...] and [i] are supported")
        i = int(key) + (self.shape[0] if key < 0 else 0)
        if not 0 <= i < self.shape[0]:
            raise IndexError(f"index {key} out of range for axis 0 of size {self.shape[0]}")
        L = self._lib.L
        fs = _check(L.H5Dget_space(self._id), "H5Dget_space")
        start = _dims((i,) + (0,) * (len(self.shape) - 1))
        count = _dims((1,) + self.shape[1:])
        _check(L.H5Sselect_hyperslab(fs, _H5S_SELECT_SET, start, None, count, None), "H5Sselect_hyperslab")
        sub = self.shape[1:]
        ms = _check(L.H5Screate_simple(len(sub), _dims(sub), None) if sub else L.H5Screate(0), "H5Screate")
        return fs, ms, sub

    def __getitem__(self, key):
        if not (key is Ellipsis or isinstance(key, (int, np.integer)) or (isinstance(key, tuple) and len(key) == 0) or
                (isinstance(key, slice) and key == slice(None))):
            return self[...][key]                # anything fancier: read the lot, let numpy index it
        fs, ms, shape = self._row(key)
        out = np.empty(shape, dtype=self.dtype)
        L = self._lib.L
        rc = L.H5Dread(self._id, self._lib.types[self.dtype], ms, fs, 0, out.ctypes.data_as(ctypes.c_void_p))
        if fs:
            L.H5Sclose(fs), L.H5Sclose(ms)
        _check(rc, f"H5Dread({self.name})")
        return out

    def __setitem__(self, key, value):
        fs, ms, shape = self._row(key)
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dtype=self.dtype), shape))
        L = self._lib.L
        rc = L.H5Dwrite(self._id, self._lib.types[self.dtype], ms, fs, 0, a.ctypes.data_as(ctypes.c_void_p))
        if fs:
            L.H5Sclose(fs), L.H5Sclose(ms)
        _check(rc, f"H5Dwrite({self.name})")

    def __array__(self, dtype=None, copy=None):
        a = self[...]
        return a if dtype is None else a.astype(dtype)

    def __len__(self):
        return self.shape[0]


class Group:
    _id = None

    def __init__(self, lib, gid, name):
        self._lib, self._id, self.name = lib, gid, name

    # -- creation ---------------------------------------------------------------------------------------
    def _parents(self, name):
        """Create the missing groups above ``name`` (h5py does this through the link-creation property list)."""
        parts = [p for p in name.strip("/").split("/") if p]
        L = self._lib.L
        for k in range(1, len(parts)):
            sub = "/".join(parts[:k]).encode()
            if L.H5Lexists(self._id, sub, 0) <= 0:
                L.H5Gclose(_check(L.H5Gcreate2(self._id, sub, 0, 0, 0), f"H5Gcreate2({sub.decode()})"))
        return "/".join(parts).encode()

    def create_group(self, name):
        path = self._parents(name)
        gid = _check(self._lib.L.H5Gcreate2(self._id, path, 0, 0, 0), f"H5Gcreate2({name})")
        return Group(self._lib, gid, name)

    def create_dataset(self, name, shape=None, dtype=None, data=None):
        if data is not None:
            data = np.asarray(data) if dtype is None else np.asarray(data, dtype=dtype)
            if data.dtype == np.bool_:
                data = data.astype("u1")
            shape = data.shape if shape is None else tuple(np.atleast_1d(shape))
            dt = data.dtype
        else:
            if shape is None:
                raise TypeError("create_dataset needs data or shape")
            shape = (int(shape),) if np.isscalar(shape) else tuple(int(s) for s in shape)
            dt = np.dtype("f8" if dtype is None else dtype)
        dt = np.dtype(dt).newbyteorder("=")
        if dt not in self._lib.types:
            raise TypeError(f"{name}: dtype {dt} is outside what this binding writes (f8 f4 i8 i4 u1)")
        path = self._parents(name)
        L = self._lib.L
        sp = _check(L.H5Screate_simple(len(shape), _dims(shape), None) if shape else L.H5Screate(0), "H5Screate")
        did = L.H5Dcreate2(self._id, path, self._lib.types[dt], sp, 0, 0, 0)
        L.H5Sclose(sp)
        ds = Dataset(self._lib, _check(did, f"H5Dcreate2({name})"), name)
        if data is not None:
            ds[...] = data
        return ds

    # -- lookup -----------------------------------------------------------------------------------------
    def __contains__(self, name):
        parts = [p for p in name.strip("/").split("/") if p]
        return all(self._lib.L.H5Lexists(self._id, "/".join(parts[:k + 1]).encode(), 0) > 0 for k in range(len(parts)))

    def __getitem__(self, name):
        if name not in self:
            raise KeyError(name)
        L = self._lib.L
        path = name.strip("/").encode()
        oid = _check(L.H5Oopen(self._id, path, 0), f"H5Oopen({name})")
        kind = L.H5Iget_type(oid)
        L.H5Oclose(oid)
        if kind == _H5I_DATASET:
            return Dataset(self._lib, _check(L.H5Dopen2(self._id, path, 0), f"H5Dopen2({name})"), name)
        if kind == _H5I_GROUP:
            return Group(self._lib, _check(L.H5Gopen2(self._id, path, 0), f"H5Gopen2({name})"), name)
        raise TypeError(f"{name}: neither a group nor a dataset")

    def keys(self):
        L = self._lib.L
        info = _GInfo()
        _check(L.H5Gget_info(self._id, ctypes.byref(info)), "H5Gget_info")
        out = []
        for i in range(info.nlinks):
            n = _check(L.H5Lget_name_by_idx(self._id, b".", _H5_INDEX_NAME, _H5_ITER_INC, i, None, 0, 0), "H5Lget_name_by_idx")
            buf = ctypes.create_string_buffer(n + 1)
            L.H5Lget_name_by_idx(self._id, b".", _H5_INDEX_NAME, _H5_ITER_INC, i, buf, n + 1, 0)
            out.append(buf.value.decode())
        return out

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def close(self):
        if self._id is not None:
            self._lib.L.H5Gclose(self._id)
            self._id = None

    def __del__(self):
        try:
            self.close()
        except Exception:            # interpreter shutdown: the library may already be gone
            pass


class File(Group):
    """``File(path, "w")`` truncates / creates, ``"r"`` opens read-only, ``"r+"`` / ``"a"`` read-write."""

    def __init__(self, path, mode="r"):
        lib = library()
        p = os.fsencode(path)
        fapl = lib.strong_close_fapl()
        if mode == "w":
            fid = lib.L.H5Fcreate(p, _H5F_ACC_TRUNC, 0, fapl)
        elif mode == "r":
            fid = lib.L.H5Fopen(p, _H5F_ACC_RDONLY, fapl)
        elif mode in ("r+", "a"):
            fid = lib.L.H5Fopen(p, _H5F_ACC_RDWR, fapl) if os.path.exists(path) else lib.L.H5Fcreate(p, _H5F_ACC_TRUNC, 0, fapl)
        else:
            if fapl:
                lib.L.H5Pclose(fapl)
            raise ValueError("mode should be w, r, r+ or a")
        if fapl:
            lib.L.H5Pclose(fapl)
        if fid < 0:
            raise OSError(f"HDF5: cannot open {path} with mode {mode}")
        super().__init__(lib, fid, "/")
        self.filename = path

    def flush(self):
        _check(self._lib.L.H5Fflush(self._id, 1), "H5Fflush")

    def close(self):
        if self._id is not None:
            self._lib.L.H5Fclose(self._id)
            self._id = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def open_file(path, mode="r"):
    """``h5py.File`` where h5py exists, this module's ``File`` otherwise (same calls for what the store uses)."""
    try:
        import h5py
        return h5py.File(path, mode)
    except ImportError:
        return File(path, mode)


def walk(group, prefix=""):
    """Depth-first (name, dataset) pairs below ``group``; works on both back ends."""
    for k in sorted(group.keys()):
        obj = group[k]
        name = f"{prefix}{k}"
        if hasattr(obj, "keys"):
            yield from walk(obj, name + "/")
        else:
            yield name, obj
