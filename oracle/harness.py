"""oracle/harness.py -- TEST INFRASTRUCTURE ONLY.

ctypes front-ends for the two CPU checkers:

  * ``port()``  -> oracle/_build/libedt_oracle.so, our plain-C restatement (oracle/edt_oracle.c)
  * ``ref()``   -> oracle/_ref/libedt_ref.so, the *real* reference compiled from
                   /root/reference/src by oracle/Makefile (strict flags);
    ``ref(fast=True)`` -> the same sources with the reference's own ``-O3 -ffast-math``.

Both expose the same numpy-level calls with the reference's Python conventions
(src/edt.pyx:639-734: C-ordered arrays are the same computation with extents and
anisotropy reversed; signed ints are reinterpreted as unsigned; bool takes the binary path).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product package never does.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

DT_U8, DT_U16, DT_U32, DT_U64, DT_F32, DT_F64, DT_BOOL = range(7)

_DTYPE_CODE = {
    np.dtype(np.uint8): DT_U8, np.dtype(np.int8): DT_U8,
    np.dtype(np.uint16): DT_U16, np.dtype(np.int16): DT_U16,
    np.dtype(np.uint32): DT_U32, np.dtype(np.int32): DT_U32,
    np.dtype(np.uint64): DT_U64, np.dtype(np.int64): DT_U64,
    np.dtype(np.float32): DT_F32, np.dtype(np.float64): DT_F64,
    np.dtype(bool): DT_BOOL,
}
_UNSIGNED = {DT_U8: np.uint8, DT_U16: np.uint16, DT_U32: np.uint32, DT_U64: np.uint64}


def build(which: str = "port") -> None:
    """Run oracle/Makefile for ``port`` and/or ``ref`` (``ref`` needs /root/reference)."""
    subprocess.run(["make", "-C", _HERE, which], check=True, capture_output=True)


def _canonical(data):
    """Mirror of the dtype/order handling in src/edt.pyx:276-289, :651-732."""
    data = np.asarray(data)
    if not data.flags.c_contiguous and not data.flags.f_contiguous:
        data = np.ascontiguousarray(data)
    order = "F" if data.flags.f_contiguous else "C"
    code = _DTYPE_CODE[data.dtype]
    if code in _UNSIGNED:
        data = data.astype(_UNSIGNED[code], order="K")  # same reinterpretation as .astype(np.uintN)
    elif code == DT_BOOL:
        data = data.view(np.uint8)
    return data, order, code


class _Lib:
    def __init__(self, path: str, prefix: str):
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} is missing -- run `make -C oracle {'port' if prefix == 'oracle' else 'ref'}`")
        self.path = path
        self.prefix = prefix
        self.lib = ctypes.CDLL(path)
        self.is_ref = prefix == "ref"

    # -- raw x-fastest calls ---------------------------------------------------------
    def raw3d(self, labels, code, sx, sy, sz, w, bb, parallel=1):
        out = np.zeros(sx * sy * sz, dtype=np.float32)
        fn = getattr(self.lib, f"{self.prefix}_edt3dsq")
        args = [labels.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(code),
                ctypes.c_int64(sx), ctypes.c_int64(sy), ctypes.c_int64(sz),
                ctypes.c_float(w[0]), ctypes.c_float(w[1]), ctypes.c_float(w[2]),
                ctypes.c_int(int(bb))]
        if self.is_ref:
            args.append(ctypes.c_int(parallel))
        args.append(out.ctypes.data_as(ctypes.c_void_p))
        rc = fn(*args)
        assert rc == 0, rc
        return out

    def raw2d(self, labels, code, sx, sy, w, bb, parallel=1):
        out = np.zeros(sx * sy, dtype=np.float32)
        fn = getattr(self.lib, f"{self.prefix}_edt2dsq")
        args = [labels.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(code),
                ctypes.c_int64(sx), ctypes.c_int64(sy),
                ctypes.c_float(w[0]), ctypes.c_float(w[1]), ctypes.c_int(int(bb))]
        if self.is_ref:
            args.append(ctypes.c_int(parallel))
        args.append(out.ctypes.data_as(ctypes.c_void_p))
        rc = fn(*args)
        assert rc == 0, rc
        return out

    def raw1d(self, labels, code, n, w, bb):
        out = np.zeros(n, dtype=np.float32)
        fn = getattr(self.lib, f"{self.prefix}_edt1dsq")
        rc = fn(labels.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(code), ctypes.c_int64(n),
                ctypes.c_float(w), ctypes.c_int(int(bb)), out.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0, rc
        return out

    # -- numpy-level API with the reference's conventions ----------------------------
    def edtsq(self, data, anisotropy=None, black_border=False, parallel=1, voxel_graph=None):
        data = np.asarray(data)
        if data.size == 0:
            return np.zeros(data.shape, dtype=np.float32)
        arr, order, code = _canonical(data)
        dims = arr.ndim
        if voxel_graph is not None:
            return self._voxel_graph(arr, order, code, voxel_graph, anisotropy, black_border)
        if dims == 1:
            w = 1.0 if anisotropy is None else float(anisotropy)
            return self.raw1d(np.ascontiguousarray(arr), code, arr.size, w, black_border)
        if dims == 2:
            a = (1.0, 1.0) if anisotropy is None else tuple(float(v) for v in anisotropy)
            if order == "F":
                sx, sy, w = arr.shape[0], arr.shape[1], (a[0], a[1])
            else:
                sx, sy, w = arr.shape[1], arr.shape[0], (a[1], a[0])
            out = self.raw2d(arr, code, sx, sy, w, black_border, parallel)
            return out.reshape(arr.shape, order=order)
        if dims == 3:
            a = (1.0, 1.0, 1.0) if anisotropy is None else tuple(float(v) for v in anisotropy)
            if order == "F":
                ext, w = arr.shape, a
            else:
                ext, w = arr.shape[::-1], a[::-1]
            out = self.raw3d(arr, code, ext[0], ext[1], ext[2], w, black_border, parallel)
            return out.reshape(arr.shape, order=order)
        raise TypeError(f"Multi-Label EDT library only supports up to 3 dimensions got {dims}.")

    def binary_edtsq(self, data, anisotropy=None, black_border=False, parallel=1):
        """pyedt::_binary_edt{2,3}dsq<T> for ANY label type (what edt::binary_edt* instantiates,
        src/edt.hpp:487-576, :681-732): labels split runs in pass 1 only.  2-D / 3-D."""
        data = np.asarray(data)
        if data.size == 0:
            return np.zeros(data.shape, dtype=np.float32)
        arr, order, code = _canonical(data)
        dims = arr.ndim
        assert dims in (2, 3)
        a = (1.0,) * dims if anisotropy is None else tuple(float(v) for v in anisotropy)
        ext = list(arr.shape if order == "F" else arr.shape[::-1])
        w = list(a if order == "F" else a[::-1])
        out = np.zeros(arr.size, dtype=np.float32)
        lp, op = arr.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)
        if self.is_ref:
            if dims == 3:
                rc = self.lib.ref_binary_edt3dsq(
                    lp, ctypes.c_int(code), ctypes.c_int64(ext[0]), ctypes.c_int64(ext[1]), ctypes.c_int64(ext[2]),
                    ctypes.c_float(w[0]), ctypes.c_float(w[1]), ctypes.c_float(w[2]), ctypes.c_int(int(black_border)),
                    ctypes.c_int(parallel), op)
            else:
                rc = self.lib.ref_binary_edt2dsq(
                    lp, ctypes.c_int(code), ctypes.c_int64(ext[0]), ctypes.c_int64(ext[1]),
                    ctypes.c_float(w[0]), ctypes.c_float(w[1]), ctypes.c_int(int(black_border)),
                    ctypes.c_int(parallel), op)
        else:
            ext += [1] * (3 - dims)
            w += [1.0] * (3 - dims)
            rc = self.lib.oracle_binary_edtsq(
                lp, ctypes.c_int(code), ctypes.c_int64(ext[0]), ctypes.c_int64(ext[1]), ctypes.c_int64(ext[2]),
                ctypes.c_float(w[0]), ctypes.c_float(w[1]), ctypes.c_float(w[2]), ctypes.c_int(int(black_border)),
                op, ctypes.c_int(dims))
        assert rc == 0, rc
        return out.reshape(arr.shape, order=order)

    def edt(self, data, anisotropy=None, black_border=False, parallel=1, voxel_graph=None):
        dt = self.edtsq(data, anisotropy, black_border, parallel, voxel_graph)
        return np.sqrt(dt, dt)

    def sdf(self, data, anisotropy=None, black_border=False, parallel=1):
        data = np.asarray(data)
        dt = self.edt(data, anisotropy, black_border, parallel)
        dt -= self.edt(data == 0, anisotropy, black_border, parallel)
        return dt

    def sdfsq(self, data, anisotropy=None, black_border=False, parallel=1):
        data = np.asarray(data)
        return (self.edtsq(data, anisotropy, black_border, parallel)
                - self.edtsq(data == 0, anisotropy, black_border, parallel))

    def _voxel_graph(self, arr, order, code, graph, anisotropy, bb):
        graph = np.asarray(graph)
        graph = np.ascontiguousarray(graph) if order == "C" else np.asfortranarray(graph)
        graph = graph.view(np.uint8) if graph.dtype.itemsize == 1 else graph.astype(np.uint8)
        dims = arr.ndim
        if dims not in (2, 3):
            raise TypeError(f"Voxel connectivity graph is only supported for 2D and 3D. Got {dims}.")
        a = (1.0,) * dims if anisotropy is None else tuple(float(v) for v in anisotropy)
        ext = arr.shape if order == "F" else arr.shape[::-1]
        w = a if order == "F" else a[::-1]
        out = np.zeros(arr.size, dtype=np.float32)
        lp = arr.ctypes.data_as(ctypes.c_void_p)
        gp = graph.ctypes.data_as(ctypes.c_void_p)
        op = out.ctypes.data_as(ctypes.c_void_p)
        if self.is_ref:
            if dims == 3:
                rc = self.lib.ref_edt3dsq_voxel_graph(
                    lp, ctypes.c_int(code), gp, ctypes.c_int64(ext[0]), ctypes.c_int64(ext[1]),
                    ctypes.c_int64(ext[2]), ctypes.c_float(w[0]), ctypes.c_float(w[1]),
                    ctypes.c_float(w[2]), ctypes.c_int(int(bb)), op)
            else:
                rc = self.lib.ref_edt2dsq_voxel_graph(
                    lp, ctypes.c_int(code), gp, ctypes.c_int64(ext[0]), ctypes.c_int64(ext[1]),
                    ctypes.c_float(w[0]), ctypes.c_float(w[1]), ctypes.c_int(int(bb)), op)
        else:
            e = list(ext) + [1] * (3 - dims)
            ww = list(w) + [1.0] * (3 - dims)
            rc = self.lib.oracle_edt3dsq_voxel_graph(
                lp, ctypes.c_int(code), gp, ctypes.c_int64(e[0]), ctypes.c_int64(e[1]),
                ctypes.c_int64(e[2]), ctypes.c_float(ww[0]), ctypes.c_float(ww[1]),
                ctypes.c_float(ww[2]), ctypes.c_int(int(bb)), op, ctypes.c_int(dims))
        assert rc == 0, rc
        return out.reshape(arr.shape, order=order)


_cache: dict = {}


def port() -> _Lib:
    if "port" not in _cache:
        _cache["port"] = _Lib(os.path.join(_HERE, "_build", "libedt_oracle.so"), "oracle")
    return _cache["port"]


def ref(fast: bool = False) -> _Lib:
    key = "ref_fast" if fast else "ref"
    if key not in _cache:
        name = "libedt_ref_fast.so" if fast else "libedt_ref.so"
        _cache[key] = _Lib(os.path.join(_HERE, "_ref", name), "ref")
    return _cache[key]


def have_ref() -> bool:
    return os.path.exists(os.path.join(_HERE, "_ref", "libedt_ref.so"))


def have_port() -> bool:
    return os.path.exists(os.path.join(_HERE, "_build", "libedt_oracle.so"))
