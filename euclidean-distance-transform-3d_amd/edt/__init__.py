"""Drop-in replacement for the reference's ``edt`` Python module, running on AMD MI355X.

Same public surface as the reference Cython binding (reference: src/edt.pyx:115-310,
:312-844): ``edt, edtsq, sdf, sdfsq, edt1d[sq], edt2d[sq], edt3d[sq]`` with the same
argument meaning, defaults and error behaviour.  The numerics run in hand-written HIP
kernels behind the C ABI of ``include/edt_hip.h``; this file is only the host-side
argument handling (shape / order / dtype dispatch), exactly as the reference keeps it in
Cython.  There is no CPU fallback: without the built library or without a GPU, calls raise.

Conventions kept from the reference:
  * arrays may be C or Fortran contiguous; C order is the same computation with extents
    and anisotropy reversed (src/edt.pyx:651-664);
  * signed integer labels are reinterpreted as unsigned (src/edt.pyx:670-705), booleans are
    one byte per voxel (:724-732);
  * ``parallel`` is accepted and ignored (the GPU grid replaces the thread pool);
  * empty input returns an empty float32 array (:281-282); >3 dims raises TypeError (:309-310).

Device-resident use (torch tensors on ``cuda``) goes through :mod:`edt.device`.
"""
from __future__ import annotations

import ctypes
import multiprocessing

import numpy as np

from . import _lib
from ._lib import EdtHipError  # noqa: F401  (re-export)

# a host that cannot run any transform fails HERE, with the reason (no built library; no GPU device node) -- _lib.probe_at_import
_lib.probe_at_import()

__all__ = [
    "edt", "edtsq", "sdf", "sdfsq",
    "edt1d", "edt1dsq", "edt2d", "edt2dsq", "edt3d", "edt3dsq",
    "each", "edt_stack", "edtsq_stack", "binary_edt", "binary_edtsq", "set_devices", "EdtHipError",
    "runs", "draw", "transfer", "erase", "reshape", "nvl",
]


def nvl(val, default_val):
    """`val`, or `default_val` when it is None (module-level helper of the reference, src/edt.pyx:115-118)."""
    return default_val if val is None else val


def reshape(arr, shape, order=None):
    """A view of a contiguous array under another shape, in the array's own memory order unless `order` ('C' / 'F') says
    otherwise -- no copy where the layout allows one (module-level helper of the reference, src/edt.pyx:851-877; an
    array that is contiguous in neither order is reshaped the numpy way)."""
    arr = np.asarray(arr)
    if order is None:
        order = "F" if arr.flags.f_contiguous else ("C" if arr.flags.c_contiguous else None)
    return arr.reshape(shape) if order is None else arr.reshape(shape, order=order)

_DTYPE_CODE = {
    np.dtype(np.uint8): _lib.U8, np.dtype(np.int8): _lib.U8,
    np.dtype(np.uint16): _lib.U16, np.dtype(np.int16): _lib.U16,
    np.dtype(np.uint32): _lib.U32, np.dtype(np.int32): _lib.U32,
    np.dtype(np.uint64): _lib.U64, np.dtype(np.int64): _lib.U64,
    np.dtype(np.float32): _lib.F32, np.dtype(np.float64): _lib.F64,
    np.dtype(bool): _lib.BOOL,
}
_UNSIGNED = {_lib.U8: np.uint8, _lib.U16: np.uint16, _lib.U32: np.uint32, _lib.U64: np.uint64}


def _label_code(data: np.ndarray) -> int:
    try:
        return _DTYPE_CODE[data.dtype]
    except KeyError:
        raise TypeError(
            f"Unsupported label dtype {data.dtype}; supported: (u)int8/16/32/64, float32, "
            "float64, bool.") from None


def _as_label_buffer(data: np.ndarray, code: int) -> np.ndarray:
    """Contiguous buffer the kernels can read: signed -> unsigned view, bool -> bytes."""
    if code in _UNSIGNED:
        want = np.dtype(_UNSIGNED[code])
        return data.view(want) if data.dtype != want else data
    if code == _lib.BOOL:
        return data.view(np.uint8)
    return data


def _ptr(arr: np.ndarray) -> ctypes.c_void_p:
    return ctypes.c_void_p(arr.ctypes.data)


# ----------------------------------------------------------------------------------------
# public API
# ----------------------------------------------------------------------------------------
def sdf(data, anisotropy=None, black_border=False, parallel=1, voxel_graph=None, order=None):
    """Signed distance function: ``edt(data) - edt(data == 0)`` (reference: src/edt.pyx:121-158)."""
    data = np.asarray(data) if isinstance(data, list) else data
    if voxel_graph is None:
        # one round trip: both transforms and the subtraction run on the device (edt_hip_sdf)
        return _transform(data, anisotropy, black_border, parallel, None, take_sqrt=True, signed=True)

    def fn(labels):
        return edt(labels, anisotropy=anisotropy, black_border=black_border, parallel=parallel,
                   voxel_graph=voxel_graph)

    dt = fn(data)
    dt -= fn(data == 0)
    return dt


def sdfsq(data, anisotropy=None, black_border=False, parallel=1, voxel_graph=None):
    """Squared signed distance function (reference: src/edt.pyx:161-202)."""
    data = np.asarray(data) if isinstance(data, list) else data
    if voxel_graph is None:
        return _transform(data, anisotropy, black_border, parallel, None, take_sqrt=False, signed=True)

    def fn(labels):
        return edtsq(labels, anisotropy=anisotropy, black_border=black_border, parallel=parallel,
                     voxel_graph=voxel_graph)

    return fn(data) - fn(data == 0)


def edt(data, anisotropy=None, black_border=False, parallel=1, voxel_graph=None, order=None):
    """Anisotropic multi-label Euclidean distance transform of a 1-D/2-D/3-D array.

    Reference: src/edt.pyx:205-242.  The square root is fused into the last GPU pass
    (correctly rounded, hence identical to the reference's ``np.sqrt``).
    """
    return _transform(data, anisotropy, black_border, parallel, voxel_graph, take_sqrt=True)


def edtsq(data, anisotropy=None, black_border=False, parallel=1, voxel_graph=None, order=None):
    """Squared distance transform (reference: src/edt.pyx:245-310)."""
    return _transform(data, anisotropy, black_border, parallel, voxel_graph, take_sqrt=False)


def edt1d(data, anisotropy=1.0, black_border=False):
    return _run(np.asarray(data), (anisotropy,), black_border, None, True, ndim=1)


def edt1dsq(data, anisotropy=1.0, black_border=False):
    return _run(np.asarray(data), (anisotropy,), black_border, None, False, ndim=1)


def edt2d(data, anisotropy=(1.0, 1.0), black_border=False, parallel=1, voxel_graph=None):
    return _run(np.asarray(data), anisotropy, black_border, voxel_graph, True, ndim=2)


def edt2dsq(data, anisotropy=(1.0, 1.0), black_border=False, parallel=1, voxel_graph=None):
    return _run(np.asarray(data), anisotropy, black_border, voxel_graph, False, ndim=2)


def edt3d(data, anisotropy=(1.0, 1.0, 1.0), black_border=False, parallel=1, voxel_graph=None):
    return _run(np.asarray(data), anisotropy, black_border, voxel_graph, True, ndim=3)


def edt3dsq(data, anisotropy=(1.0, 1.0, 1.0), black_border=False, parallel=1, voxel_graph=None):
    return _run(np.asarray(data), anisotropy, black_border, voxel_graph, False, ndim=3)


def binary_edtsq(data, anisotropy=None, black_border=False, parallel=1):
    """The C++ facade's ``edt::binary_edtsq`` for 2-D / 3-D arrays of ANY label type (reference:
    src/edt.hpp:895-951 -> ``pyedt::_binary_edt{2,3}dsq<T>``, :487-576, :681-755): labels split runs along x only;
    along y and z every non-zero voxel is one foreground.  Equal to ``edtsq`` on 0/1 input.  (The reference's Python
    module reaches this route for ``bool`` arrays only; exposed here so the facade's semantics can be tested.)"""
    data = np.asarray(data)
    if data.ndim not in (2, 3):
        raise TypeError("binary_edtsq: 2-D or 3-D arrays")
    an = (1.0,) * data.ndim if anisotropy is None else anisotropy
    return _run(data, an, black_border, None, False, ndim=data.ndim, binary=True)


def binary_edt(data, anisotropy=None, black_border=False, parallel=1):
    data = np.asarray(data)
    if data.ndim not in (2, 3):
        raise TypeError("binary_edt: 2-D or 3-D arrays")
    an = (1.0,) * data.ndim if anisotropy is None else anisotropy
    return _run(data, an, black_border, None, True, ndim=data.ndim, binary=True)


def set_devices(devices=None):
    """Z-shard every 3-D transform of host arrays over these GPUs of THIS process (``edt_hip_set_devices``: a host
    thread per device, one peer-to-peer exchange over xGMI, see ``csrc/edt_multi.hip``); ``None`` / ``[]`` = back to
    the current device alone.  ``EDT_HIP_DEVICES=0,1,...`` in the environment presets the list.  Results are
    bit-identical either way."""
    devs = [] if devices is None else [int(d) for d in devices]
    arr = (ctypes.c_int * max(1, len(devs)))(*devs)
    _lib.check(_lib.load().edt_hip_set_devices(ctypes.cast(arr, ctypes.c_void_p), len(devs)))


def _stack(images, anisotropy, black_border, take_sqrt):
    images = np.asarray(images)
    if images.ndim != 3:
        raise TypeError("a stack of 2-D images is a 3-D array (count, height, width)")
    if images.size == 0:
        return np.zeros(images.shape, dtype=np.float32)
    images = np.ascontiguousarray(images)      # C order: the last axis (width) is x
    code = _label_code(images)
    buf = _as_label_buffer(images, code)
    an = (1.0, 1.0) if anisotropy is None else tuple(float(np.float32(a)) for a in anisotropy)
    if len(an) != 2:
        raise ValueError("anisotropy of a 2-D image has 2 entries")
    count, sy, sx = images.shape
    out = np.empty(images.shape, dtype=np.float32)
    _lib.check(_lib.load().edt_hip_edt2dsq_batch(_ptr(buf), code, sx, sy, count, an[1], an[0],
                                                 1 if black_border else 0, 1 if take_sqrt else 0, _ptr(out)))
    return out


def edtsq_stack(images, anisotropy=None, black_border=False):
    """Squared EDT of every 2-D image of ``images[count, height, width]`` independently, in one call (an
    extension over the reference's API: the stack goes through the GPU as one batch, edt_hip_edt2dsq_batch).
    ``anisotropy`` = (height spacing, width spacing), as for a C-ordered 2-D array."""
    return _stack(images, anisotropy, black_border, take_sqrt=False)


def edt_stack(images, anisotropy=None, black_border=False):
    return _stack(images, anisotropy, black_border, take_sqrt=True)


# ----------------------------------------------------------------------------------------
# argument handling (mirrors src/edt.pyx:276-310) and dispatch into the C ABI
# ----------------------------------------------------------------------------------------
def _transform(data, anisotropy, black_border, parallel, voxel_graph, take_sqrt, signed=False):
    if isinstance(data, list):
        data = np.array(data)
    data = np.asarray(data)
    dims = data.ndim

    if data.size == 0:
        return np.zeros(shape=data.shape, dtype=np.float32)

    if parallel is not None and parallel <= 0:
        parallel = multiprocessing.cpu_count()  # accepted, unused

    if voxel_graph is not None and dims not in (2, 3):
        raise TypeError(
            "Voxel connectivity graph is only supported for 2D and 3D. Got {}.".format(dims))

    if dims == 1:
        anisotropy = (1.0 if anisotropy is None else anisotropy,)
    elif dims == 2:
        anisotropy = (1.0, 1.0) if anisotropy is None else anisotropy
    elif dims == 3:
        anisotropy = (1.0, 1.0, 1.0) if anisotropy is None else anisotropy
    else:
        raise TypeError(
            "Multi-Label EDT library only supports up to 3 dimensions got {}.".format(dims))
    return _run(data, anisotropy, black_border, voxel_graph, take_sqrt, ndim=dims, signed=signed)


def _run(data, anisotropy, black_border, voxel_graph, take_sqrt, ndim, signed=False, binary=False):
    if data.ndim != ndim:
        raise TypeError(f"expected a {ndim}-D array, got {data.ndim}-D")
    if data.size == 0:
        return np.zeros(shape=data.shape, dtype=np.float32)
    if not data.flags.c_contiguous and not data.flags.f_contiguous:
        data = np.ascontiguousarray(data)
    order = "F" if data.flags.f_contiguous else "C"
    code = _label_code(data)
    buf = _as_label_buffer(data, code)

    if np.ndim(anisotropy) == 0:
        anisotropy = (anisotropy,) * ndim if ndim == 1 else anisotropy
    weights = tuple(float(np.float32(a)) for a in np.asarray(anisotropy, dtype=np.float64).reshape(-1))
    if len(weights) != ndim:
        raise ValueError(f"anisotropy must have {ndim} entries, got {len(weights)}")
    # (stated deviation: the reference does not validate voxel sizes.  Along the fastest axis a size enters pass 1 as itself -- a
    # negative one makes the reference's backward sweep cross label boundaries, src/edt.hpp:107-109 --, along the other axes only
    # as its square, src/edt.hpp:181, :258: the sign is meaningless there and accepted, as the reference does; zero, NaN and
    # inf are refused everywhere, here and at the C ABI, include/edt_hip.h)
    fastest = 0 if order == "F" else ndim - 1
    if not all(np.isfinite(a) and a != 0.0 for a in weights) or weights[fastest] < 0.0:
        raise ValueError(f"anisotropy must be finite and non-zero (and positive along the fastest axis), got {weights}")

    # x is the fastest axis of the buffer the kernels see.
    if order == "F":
        extents, w = tuple(data.shape), weights
    else:
        extents, w = tuple(data.shape[::-1]), weights[::-1]

    lib = _lib.load()
    out = np.empty(data.size, dtype=np.float32)
    bb = 1 if black_border else 0

    if voxel_graph is not None:
        if ndim not in (2, 3):
            raise TypeError(
                "Voxel connectivity graph is only supported for 2D and 3D. Got {}.".format(ndim))
        graph = np.asarray(voxel_graph)
        if graph.shape != data.shape:
            raise ValueError("voxel_graph must have the same shape as data")
        graph = np.ascontiguousarray(graph) if order == "C" else np.asfortranarray(graph)
        # only the low 6 bits are meaningful (src/edt.pyx:748-752)
        graph = graph.view(np.uint8) if graph.dtype.itemsize == 1 else graph.astype(np.uint8, order="K")
        if ndim == 2:
            rc = lib.edt_hip_edt2dsq_voxel_graph(_ptr(buf), code, _ptr(graph), extents[0], extents[1],
                                                 w[0], w[1], bb, _ptr(out))
        else:
            rc = lib.edt_hip_edt3dsq_voxel_graph(_ptr(buf), code, _ptr(graph), extents[0], extents[1],
                                                 extents[2], w[0], w[1], w[2], bb, _ptr(out))
        _lib.check(rc)
        if take_sqrt:
            np.sqrt(out, out)
        return out.reshape(data.shape, order=order)

    if signed:  # sdf / sdfsq: edt(x) - edt(x == 0), everything but the two copies on the device
        e = tuple(extents) + (1,) * (3 - ndim)
        ww = tuple(w) + (1.0,) * (3 - ndim)
        _lib.check(lib.edt_hip_sdf(_ptr(buf), code, ndim, e[0], e[1], e[2], ww[0], ww[1], ww[2], bb,
                                   0 if take_sqrt else 1, _ptr(out)))
    elif binary:
        e = tuple(extents) + (1,) * (3 - ndim)
        ww = tuple(w) + (1.0,) * (3 - ndim)
        _lib.check(lib.edt_hip_binary_edtsq(_ptr(buf), code, ndim, e[0], e[1], e[2], ww[0], ww[1], ww[2], bb,
                                            1 if take_sqrt else 0, _ptr(out)))
    elif ndim == 1:
        rc = lib.edt_hip_squared_edt_1d_multi_seg(_ptr(buf), code, _ptr(out), data.size, 1, w[0], bb)
        _lib.check(rc)
        if take_sqrt:
            np.sqrt(out, out)
    elif ndim == 2:
        fn = lib.edt_hip_edt2d if take_sqrt else lib.edt_hip_edt2dsq
        _lib.check(fn(_ptr(buf), code, extents[0], extents[1], w[0], w[1], bb, 1, _ptr(out)))
    else:
        fn = lib.edt_hip_edt3d if take_sqrt else lib.edt_hip_edt3dsq
        _lib.check(fn(_ptr(buf), code, extents[0], extents[1], extents[2], w[0], w[1], w[2], bb, 1,
                      _ptr(out)))
    return out.reshape(data.shape, order=order)


# ----------------------------------------------------------------------------------------
# Run utilities and each(): per-label views of a distance transform (reference:
# src/edt.pyx:847-994, src/edt_voxel_graph.hpp:238-310).  Host-side conveniences on top of the DT
# with the reference's semantics (runs are half-open [start, end) ranges of the array's memory
# order); the device-resident counterpart of each() is edt.device.each / select_label.
# ----------------------------------------------------------------------------------------
def _flat(arr: np.ndarray) -> np.ndarray:
    """1-D view in memory order (src/edt.pyx:851-877: in-place reshape of a contiguous array)."""
    if arr.flags.f_contiguous and not arr.flags.c_contiguous:
        return arr.reshape(-1, order="F")
    return arr.reshape(-1)


def runs(labels):
    """``{label: [(start, end), ...]}`` -- where each label lies (src/edt_voxel_graph.hpp:238-268)."""
    flat = _flat(np.asarray(labels))
    out = {}
    if flat.size == 0:
        return out
    cuts = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    starts = np.concatenate(([0], cuts))
    ends = np.concatenate((cuts, [flat.size]))
    for s, e, v in zip(starts.tolist(), ends.tolist(), flat[starts].tolist()):
        out.setdefault(v, []).append((s, e))
    return dict(sorted(out.items()))  # std::map iterates in key order


def _check_runs(rns, voxels):
    for s, e in rns:
        if s < 0 or e > voxels or e < 0 or s >= e:
            raise RuntimeError("Invalid run.")  # src/edt_voxel_graph.hpp:277-283


def draw(label, runs, image):
    """Write ``label`` into ``image`` along ``runs`` (set_run_voxels, src/edt_voxel_graph.hpp:270-288).
    Like the reference (src/edt.pyx:895-913) it returns the FLAT view of the image it wrote through."""
    flat = _flat(image)
    _check_runs(runs, flat.size)
    for s, e in runs:
        flat[s:e] = label
    return flat


def transfer(runs, src, dest):
    """Copy ``src`` to ``dest`` along ``runs`` (transfer_run_voxels, src/edt_voxel_graph.hpp:290-310);
    returns the flat view of ``dest`` (src/edt.pyx:915-935)."""
    fs, fd = _flat(src), _flat(dest)
    assert fs.size == fd.size
    _check_runs(runs, fs.size)
    for s, e in runs:
        fd[s:e] = fs[s:e]
    return fd


def erase(runs, image):
    """Zero ``image`` along ``runs`` (src/edt.pyx:937-947)."""
    return draw(0, runs, image)


class _PerLabelImages:
    """Sized iterable behind :func:`each`: one ``(label, image)`` pair per non-zero label, in ascending
    label order, where ``image`` is ``dt`` on the label's voxels and 0 elsewhere.

    The label volume is scanned ONCE into a run table (start, end, value per run, sorted by value); every
    image is then painted run by run.  With ``reuse`` one canvas serves all labels: it is handed out
    read-only and wiped along the same runs before the next label is painted."""

    def __init__(self, labels, dt, reuse):
        self._shape = labels.shape
        self._order = "F" if labels.flags.f_contiguous else "C"
        self._dt = _flat(dt)
        self._reuse = reuse
        flat = _flat(labels)
        if flat.size == 0:
            cuts = np.zeros(0, dtype=np.int64)
            starts = ends = cuts
        else:
            cuts = np.flatnonzero(flat[1:] != flat[:-1]) + 1
            starts = np.concatenate(([0], cuts))
            ends = np.concatenate((cuts, [flat.size]))
        values = flat[starts]
        keep = values != 0
        starts, ends, values = starts[keep], ends[keep], values[keep]
        by_value = np.argsort(values, kind="stable")           # runs of one label stay in memory order
        self._starts, self._ends = starts[by_value], ends[by_value]
        self._labels, first = np.unique(values[by_value], return_index=True)
        self._bounds = np.append(first, values.size)            # runs of label i: [bounds[i], bounds[i+1])

    def __len__(self):
        return int(self._labels.size)

    def _paint(self, canvas, i, source):
        for s, e in zip(self._starts[self._bounds[i]:self._bounds[i + 1]].tolist(),
                        self._ends[self._bounds[i]:self._bounds[i + 1]].tolist()):
            canvas[s:e] = source[s:e] if source is not None else 0

    def __iter__(self):
        image = flat_image = None
        for i, label in enumerate(self._labels.tolist()):
            if image is None or not self._reuse:
                image = np.zeros(self._shape, dtype=np.float32, order=self._order)
                flat_image = _flat(image)
            self._paint(flat_image, i, self._dt)
            if not self._reuse:
                yield label, image
                continue
            image.flags.writeable = False
            yield label, image
            image.flags.writeable = True
            self._paint(flat_image, i, None)


def each(labels, dt, in_place=False):
    """Iterate ``(label, image)``: the distance transform ``dt`` restricted to each non-zero label of
    ``labels`` in turn (same contract as the reference's ``edt.each``, src/edt.pyx:950-994: ``len()`` is the
    number of labels, ``in_place=True`` reuses ONE read-only image instead of allocating one per label).
    For device-resident data see :func:`edt.device.each`."""
    return _PerLabelImages(np.asarray(labels), np.asarray(dt), bool(in_place))
