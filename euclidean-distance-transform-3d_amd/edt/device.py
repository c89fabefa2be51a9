"""Device-resident entry points: torch tensors on an AMD GPU in, torch tensors out.

torch is used only as plumbing (device memory, streams); all arithmetic happens in the HIP
kernels behind ``edt_hip_edtsq_device`` (include/edt_hip.h).  Nothing is copied to the host
and nothing is allocated per call once a :class:`Plan` exists, so a call only enqueues
kernels on the current stream (graph-capturable, timeable with events).

Tensor layout: a contiguous tensor of shape ``(a, b, c)`` has its LAST dimension fastest, so
it is the same computation as a C-ordered numpy array: extents and anisotropy are reversed
before they reach the kernels (reference: src/edt.pyx:651-656).
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib

_TORCH_CODE = {
    torch.uint8: _lib.U8, torch.int8: _lib.U8,
    torch.int16: _lib.U16, torch.int32: _lib.U32, torch.int64: _lib.U64,
    torch.float32: _lib.F32, torch.float64: _lib.F64, torch.bool: _lib.BOOL,
}
for _name, _code in (("uint16", _lib.U16), ("uint32", _lib.U32), ("uint64", _lib.U64)):
    if hasattr(torch, _name):
        _TORCH_CODE[getattr(torch, _name)] = _code


def dtype_code(dtype: torch.dtype) -> int:
    try:
        return _TORCH_CODE[dtype]
    except KeyError:
        raise TypeError(f"Unsupported label dtype {dtype}") from None


def as_device_tensor(obj) -> torch.Tensor:
    """A zero-copy torch view of any device array: a torch tensor itself, an object that speaks DLPack (``__dlpack__``:
    CuPy, JAX, recent Numba ...) or the CUDA array interface (``__cuda_array_interface__``: Numba device arrays, CuPy --
    also what those libraries expose on ROCm).  Every entry point of this module takes its array arguments through here
    (the boundary of src/edt.pyx:639-734, for callers whose data never was a numpy array); results are torch tensors,
    which speak both protocols in the other direction."""
    if isinstance(obj, torch.Tensor):
        t = obj
    elif hasattr(obj, "__dlpack__"):
        t = torch.from_dlpack(obj)
    elif hasattr(obj, "__cuda_array_interface__"):
        t = torch.as_tensor(obj, device=torch.device("cuda", torch.cuda.current_device()))
    else:
        raise TypeError(f"expected a device array (torch tensor, __dlpack__ or __cuda_array_interface__), got {type(obj).__name__}")
    if not t.is_cuda:
        raise TypeError("edt.device takes device-resident arrays; host arrays go through edt.edtsq / edt.edt / edt.sdf")
    return t


def _stream_ptr() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class Plan:
    """Pre-sized scratch for volumes of one shape/dtype (x-fastest extents ``(sx, sy, sz)``)."""

    def __init__(self, extents_xyz, code: int, device=None, small_workspace=False):
        """small_workspace: scratch = the four bit planes only (EDT_FLAG_SMALL_WORKSPACE: passes X and Y exchange
        fp32 values instead of 16-bit distance indices -- no index slab of up to 256 MiB, a few per cent slower)"""
        self.lib = _lib.load()
        ext = tuple(int(e) for e in extents_xyz)
        self.ndim = len(ext)
        self.ext = ext + (1,) * (3 - self.ndim)
        self.code = code
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.base_flags = _lib.FLAG_SMALL_WORKSPACE if small_workspace else 0
        nbytes = self.lib.edt_hip_workspace_bytes_flags(code, self.ndim, *self.ext, self.base_flags)
        if nbytes == 0:
            _lib.check(-2)
        self.workspace = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        self._generic_ws = None  # the size-agnostic kernels need a second volume: allocated on first use
        self.voxels = int(np.prod(self.ext))

    def _workspace_for(self, flags):
        if not (flags & _lib.FLAG_FORCE_GENERIC):
            return self.workspace
        if self._generic_ws is None:
            nbytes = self.lib.edt_hip_workspace_bytes_flags(self.code, self.ndim, *self.ext, flags)
            self._generic_ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self._generic_ws

    def run(self, labels: torch.Tensor, weights_xyz, black_border=False, sqrt=False,
            out: torch.Tensor | None = None, force_generic=False, batch2d=False, binary=False,
            signed=False) -> torch.Tensor:
        """Enqueue the transform of ``labels`` (device, contiguous, ``voxels`` elements).  binary: the reference's
        binary route for multi-valued labels (EDT_FLAG_BINARY_YZ: labels split runs along x only).  signed: the signed
        transform (EDT_FLAG_SIGNED: sdf / sdfsq as ONE transform; :meth:`signed_supported` says whether the shape is served)."""
        if not labels.is_cuda or not labels.is_contiguous():
            raise ValueError("labels must be a contiguous device tensor")
        if labels.numel() != self.voxels:
            raise ValueError("labels size does not match the plan")
        # (same-width integer views are how uint32 / uint64 labels reach torch builds without those dtypes)
        if labels.element_size() != _lib.DTYPE_SIZE[self.code] or (
                labels.dtype.is_floating_point != (self.code in (_lib.F32, _lib.F64))):
            raise TypeError(f"labels dtype {labels.dtype} does not match the plan's dtype code {self.code}")
        if out is None:
            out = torch.empty(labels.shape, dtype=torch.float32, device=labels.device)
        w = tuple(float(np.float32(v)) for v in weights_xyz) + (1.0,) * (3 - self.ndim)
        flags = ((_lib.FLAG_BLACK_BORDER if black_border else 0) | (_lib.FLAG_SQRT if sqrt else 0)
                 | (_lib.FLAG_FORCE_GENERIC if force_generic else 0)
                 | (_lib.FLAG_BATCH_2D if batch2d else 0) | (_lib.FLAG_BINARY_YZ if binary else 0)
                 | (_lib.FLAG_SIGNED if signed else 0) | self.base_flags)
        ws = self._workspace_for(flags)
        rc = self.lib.edt_hip_edtsq_device(
            ctypes.c_void_p(labels.data_ptr()), self.code, self.ndim, *self.ext, w[0], w[1], w[2],
            flags, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream_ptr())
        _lib.check(rc)
        return out


    def signed_supported(self) -> bool:
        return bool(self.lib.edt_hip_signed_supported(self.code, self.ndim, *self.ext, self.base_flags))


_plans: dict = {}


def _plan_for(ext, code, device) -> Plan:
    # one plan (= one scratch buffer) per stream: transforms of one shape issued on different streams
    # must not share scratch, and a workspace is only ever used on the stream it was allocated on
    key = (tuple(ext), code, str(device), torch.cuda.current_stream(device).cuda_stream)
    if key not in _plans:
        if len(_plans) > 8:
            _plans.clear()
        _plans[key] = Plan(ext, code, device)
    return _plans[key]


def _transform(labels, anisotropy, black_border, sqrt, force_generic=False, binary=False, signed=False):
    """signed: the signed transform where the shape is served (one transform), None returned where it is not"""
    labels = as_device_tensor(labels)
    if labels.numel() == 0:
        return torch.zeros(labels.shape, dtype=torch.float32, device=labels.device)
    if labels.dim() < 1 or labels.dim() > 3:
        raise TypeError(
            "Multi-Label EDT library only supports up to 3 dimensions got {}.".format(labels.dim()))
    labels = labels.contiguous()
    nd = labels.dim()
    an = (1.0,) * nd if anisotropy is None else (
        (float(anisotropy),) if np.ndim(anisotropy) == 0 else tuple(float(a) for a in anisotropy))
    ext = tuple(labels.shape[::-1])
    w = an[::-1]
    plan = _plan_for(ext, dtype_code(labels.dtype), labels.device)
    if signed and not plan.signed_supported():
        return None
    return plan.run(labels, w, black_border, sqrt, force_generic=force_generic, binary=binary, signed=signed)


def binary_edtsq(labels: torch.Tensor, anisotropy=None, black_border=False) -> torch.Tensor:
    """``edt::binary_edtsq`` on a 2-D / 3-D device tensor of ANY label type (reference: src/edt.hpp:487-576, :681-755):
    labels split runs along x only; along y and z every non-zero voxel is one foreground."""
    labels = as_device_tensor(labels)
    if labels.dim() not in (2, 3):
        raise TypeError("binary_edtsq: 2-D or 3-D tensors")
    return _transform(labels, anisotropy, black_border, sqrt=False, binary=True)


def edtsq(labels: torch.Tensor, anisotropy=None, black_border=False) -> torch.Tensor:
    """Squared EDT of a device tensor (same semantics as :func:`edt.edtsq` on a C-ordered array)."""
    return _transform(labels, anisotropy, black_border, sqrt=False)


def edt(labels: torch.Tensor, anisotropy=None, black_border=False) -> torch.Tensor:
    return _transform(labels, anisotropy, black_border, sqrt=True)


def _stack2d(images, anisotropy, black_border, sqrt):
    images = as_device_tensor(images)
    if images.dim() != 3:
        raise TypeError("a stack of 2-D images is a 3-D tensor (count, height, width)")
    if images.numel() == 0:
        return torch.zeros(images.shape, dtype=torch.float32, device=images.device)
    images = images.contiguous()
    an = (1.0, 1.0) if anisotropy is None else tuple(float(a) for a in anisotropy)
    if len(an) != 2:
        raise ValueError("anisotropy of a 2-D image has 2 entries")
    ext = tuple(images.shape[::-1])            # (width, height, count): x fastest
    plan = _plan_for(ext, dtype_code(images.dtype), images.device)
    return plan.run(images, (an[1], an[0], 1.0), black_border, sqrt, batch2d=True)


def edtsq_stack(images: torch.Tensor, anisotropy=None, black_border=False) -> torch.Tensor:
    """Squared EDT of every image of a stack ``(count, height, width)`` independently -- one launch per pass for
    the whole stack (EDT_FLAG_BATCH_2D), so that many small images fill the chip.  Same result as calling
    :func:`edtsq` on each image."""
    return _stack2d(images, anisotropy, black_border, sqrt=False)


def edt_stack(images: torch.Tensor, anisotropy=None, black_border=False) -> torch.Tensor:
    return _stack2d(images, anisotropy, black_border, sqrt=True)


def sdf(labels: torch.Tensor, anisotropy=None, black_border=False) -> torch.Tensor:
    """``edt(x) - edt(x == 0)`` without leaving the device (reference: src/edt.pyx:148-158) -- as ONE transform where the
    shape allows (EDT_FLAG_SIGNED: label 0 measured like every label, its voxels negated; bit-identical to the definition,
    whose two fields have disjoint supports), else as the two transforms and the subtraction."""
    return _signed(labels, anisotropy, black_border, sqrt=True)


def sdfsq(labels: torch.Tensor, anisotropy=None, black_border=False) -> torch.Tensor:
    """``edtsq(x) - edtsq(x == 0)`` without leaving the device (reference: src/edt.pyx:161-202)."""
    return _signed(labels, anisotropy, black_border, sqrt=False)


def _signed(labels, anisotropy, black_border, sqrt, one_transform=True):
    lib = _lib.load()
    labels = as_device_tensor(labels).contiguous()
    if labels.numel() == 0:
        return torch.zeros(labels.shape, dtype=torch.float32, device=labels.device)
    if one_transform and 2 <= labels.dim() <= 3:
        dt = _transform(labels, anisotropy, black_border, sqrt=sqrt, signed=True)
        if dt is not None:
            return dt
    dt = _transform(labels, anisotropy, black_border, sqrt=sqrt)
    mask = torch.empty(labels.shape, dtype=torch.bool, device=labels.device)
    _lib.check(lib.edt_hip_is_background_device(
        ctypes.c_void_p(labels.data_ptr()), dtype_code(labels.dtype),
        ctypes.c_void_p(mask.data_ptr()), labels.numel(), _stream_ptr()))
    bg = _transform(mask, anisotropy, black_border, sqrt=sqrt)
    _lib.check(lib.edt_hip_subtract_device(
        ctypes.c_void_p(dt.data_ptr()), ctypes.c_void_p(bg.data_ptr()),
        ctypes.c_void_p(dt.data_ptr()), dt.numel(), _stream_ptr()))
    return dt


def edtsq_voxel_graph(labels: torch.Tensor, voxel_graph: torch.Tensor, anisotropy=None,
                      black_border=False) -> torch.Tensor:
    """Squared EDT with a voxel connectivity graph (reference: edt.edtsq(..., voxel_graph=g),
    src/edt.pyx:736-845) on device tensors: a 2-D or 3-D labels tensor and a uint8 graph of the same shape
    (C order; bit layout of the reference).  Everything stays in HBM (edt_hip_edtsq_voxel_graph_device)."""
    labels, voxel_graph = as_device_tensor(labels), as_device_tensor(voxel_graph)
    if labels.dim() not in (2, 3) or voxel_graph.shape != labels.shape or voxel_graph.dtype != torch.uint8:
        raise TypeError("voxel_graph needs a 2-D or 3-D volume and a uint8 graph of the same shape")
    lib = _lib.load()
    labels, voxel_graph = labels.contiguous(), voxel_graph.contiguous()
    ndim = labels.dim()
    an = (1.0,) * ndim if anisotropy is None else tuple(float(a) for a in anisotropy)
    # a C-ordered tensor (z, y, x) is the x-fastest volume with extents and weights reversed
    ext = tuple(labels.shape[::-1]) + (1,) * (3 - ndim)
    w = an[::-1] + (1.0,) * (3 - ndim)
    out = torch.empty(labels.shape, dtype=torch.float32, device=labels.device)
    nbytes = lib.edt_hip_voxel_graph_workspace_bytes(ndim, ext[0], ext[1], ext[2])
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=labels.device)
    _lib.check(lib.edt_hip_edtsq_voxel_graph_device(
        ctypes.c_void_p(labels.data_ptr()), dtype_code(labels.dtype), ctypes.c_void_p(voxel_graph.data_ptr()),
        ndim, ext[0], ext[1], ext[2], w[0], w[1], w[2], _lib.FLAG_BLACK_BORDER if black_border else 0,
        ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream_ptr()))
    return out


_NP_OF_CODE = {_lib.U8: np.uint8, _lib.BOOL: np.uint8, _lib.U16: np.uint16, _lib.U32: np.uint32,
               _lib.U64: np.uint64, _lib.F32: np.float32, _lib.F64: np.float64}


def select_label(labels: torch.Tensor, dt: torch.Tensor, key, out: torch.Tensor = None) -> torch.Tensor:
    """``dt`` where ``labels == key``, 0 elsewhere -- the image :func:`edt.each` yields for one label,
    as one streaming kernel (edt_hip_select_label_device)."""
    labels, dt = as_device_tensor(labels), as_device_tensor(dt)
    if labels.shape != dt.shape or dt.dtype != torch.float32:
        raise ValueError("dt must be a float32 tensor of the labels' shape")
    labels, dt = labels.contiguous(), dt.contiguous()
    if out is None:
        out = torch.empty_like(dt)
    code = dtype_code(labels.dtype)
    # the key in the labels' own representation (signed labels are compared as their bit patterns)
    if labels.dtype in (torch.int8, torch.int16, torch.int32, torch.int64):
        signed = {torch.int8: np.int8, torch.int16: np.int16, torch.int32: np.int32, torch.int64: np.int64}
        host = np.array([key], dtype=signed[labels.dtype]).view(_NP_OF_CODE[code])
    else:
        host = np.array([key], dtype=_NP_OF_CODE[code])
    _lib.check(_lib.load().edt_hip_select_label_device(
        ctypes.c_void_p(labels.data_ptr()), code, ctypes.c_void_p(dt.data_ptr()),
        ctypes.c_void_p(host.ctypes.data), ctypes.c_void_p(out.data_ptr()), labels.numel(), _stream_ptr()))
    return out


def runs(labels: torch.Tensor):
    """Device-side ``extract_runs`` (reference: src/edt_voxel_graph.hpp:238-268): the maximal constant runs of
    the flattened (C-order) tensor as three device tensors ``(starts, ends, values)``, in memory order
    (run k covers ``[starts[k], ends[k])`` and holds ``values[k]``).  Two enqueue-only kernels sweeps
    (edt_hip_extract_runs_device); the only host synchronisation is reading the run count."""
    lib = _lib.load()
    labels = as_device_tensor(labels).contiguous()
    n = labels.numel()
    dev = labels.device
    if n == 0:
        z = torch.zeros(0, dtype=torch.int64, device=dev)
        return z, z.clone(), labels.reshape(-1)
    code = dtype_code(labels.dtype)
    ws = torch.empty(int(lib.edt_hip_runs_workspace_bytes(n)), dtype=torch.uint8, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    lp, cp, wp = ctypes.c_void_p(labels.data_ptr()), ctypes.c_void_p(count.data_ptr()), ctypes.c_void_p(ws.data_ptr())
    _lib.check(lib.edt_hip_extract_runs_device(lp, code, n, None, 0, cp, wp, ws.numel(), _stream_ptr()))
    nruns = int(count.item())
    starts = torch.empty(nruns, dtype=torch.int64, device=dev)
    _lib.check(lib.edt_hip_extract_runs_device(lp, code, n, ctypes.c_void_p(starts.data_ptr()), nruns, cp, wp,
                                               ws.numel(), _stream_ptr()))
    ends = torch.empty_like(starts)
    ends[:-1] = starts[1:]
    ends[-1] = n
    return starts, ends, labels.reshape(-1)[starts]


class _DeviceLabelImages:
    """Sized iterable behind :func:`each`: the run table of the labels is extracted once on the device; every
    label's image is then produced by ONE streaming kernel over just the span of memory its runs cover
    (first run start .. last run end) instead of the whole volume."""

    def __init__(self, labels, dt, reuse):
        labels, dt = as_device_tensor(labels), as_device_tensor(dt)
        if labels.shape != dt.shape or dt.dtype != torch.float32:
            raise ValueError("dt must be a float32 tensor of the labels' shape")
        self.labels, self.dt, self.reuse = labels.contiguous(), dt.contiguous(), reuse
        starts, ends, values = runs(self.labels)
        keep = values != 0
        starts, ends, values = starts[keep], ends[keep], values[keep]
        self.keys, inverse = torch.unique(values, return_inverse=True)   # over RUNS, not voxels
        nk = self.keys.numel()
        big = self.labels.numel()
        lo = torch.full((nk,), big, dtype=torch.int64, device=labels.device).scatter_reduce(0, inverse, starts, "amin")
        hi = torch.zeros((nk,), dtype=torch.int64, device=labels.device).scatter_reduce(0, inverse, ends, "amax")
        self._keys, self._lo, self._hi = self.keys.tolist(), lo.tolist(), hi.tolist()

    def __len__(self):
        return len(self._keys)

    def _select(self, key, lo, hi, out):
        lab, dt, flat = self.labels.reshape(-1), self.dt.reshape(-1), out.reshape(-1)
        select_label(lab[lo:hi], dt[lo:hi], key, out=flat[lo:hi])

    def __iter__(self):
        shared = torch.zeros_like(self.dt) if self.reuse else None
        prev = None
        for key, lo, hi in zip(self._keys, self._lo, self._hi):
            if self.reuse:
                if prev is not None:
                    shared.reshape(-1)[prev[0]:prev[1]].zero_()   # wipe only what the previous label wrote
                out = shared
                prev = (lo, hi)
            else:
                out = torch.zeros_like(self.dt)
            self._select(key, lo, hi, out)
            yield key, out


def each(labels: torch.Tensor, dt: torch.Tensor, in_place: bool = False):
    """Device-resident :func:`edt.each` (reference: src/edt.pyx:950-994): an iterable of
    ``(label, image)`` with ``image = dt`` restricted to that label, zeros elsewhere; the label 0 is
    skipped.  ``in_place=True`` reuses ONE output tensor for every label (the reference's read-only
    in-place image).  The DT never leaves the device."""
    return _DeviceLabelImages(labels, dt, bool(in_place))


def pass_times():
    """Durations (ms) of the kernels of the last profiled call, as ``[(name, ms), ...]``."""
    lib = _lib.load()
    cap = 256  # (the sharded phases log several calls per step)
    buf = (ctypes.c_float * cap)()
    n = lib.edt_hip_get_pass_times(ctypes.cast(buf, ctypes.c_void_p), cap)
    return [(lib.edt_hip_get_pass_name(i).decode(), float(buf[i])) for i in range(min(n, cap))]


def set_profiling(enabled: bool) -> None:
    _lib.check(_lib.load().edt_hip_set_profiling(1 if enabled else 0))
