"""GPU tier: ONE process, SEVERAL devices behind the C ABI (csrc/edt_multi.hip: a host thread per device, slab
records, one peer-to-peer exchange).  The box has one GPU, so the device list repeats ordinal 0 ("virtual
devices"): every code path of the driver runs -- partition, one-slice halo, per-destination records, the exchange as
hipMemcpyPeerAsync, the Z pass per Y-slab, the strided copy back -- only the transfers stay on one device."""
import ctypes
import os

import numpy as np
import pytest

from synth import voronoi_labels

pytestmark = pytest.mark.gpu


def _multi(lab, an, bb, sqrt, devices):
    from edt import _lib
    lib = _lib.load()
    lab = np.asfortranarray(lab)
    out = np.empty(lab.size, dtype=np.float32)
    devs = (ctypes.c_int * len(devices))(*devices)
    code = {1: _lib.U8, 2: _lib.U16, 4: _lib.U32, 8: _lib.U64}[lab.dtype.itemsize]
    rc = lib.edt_hip_edt3dsq_multi(lab.ctypes.data, code, lab.shape[0], lab.shape[1], lab.shape[2], an[0], an[1], an[2],
                                   int(bb), int(sqrt), out.ctypes.data, ctypes.cast(devs, ctypes.c_void_p), len(devices))
    _lib.check(rc)
    return out.reshape(lab.shape, order="F")


@pytest.mark.parametrize("ndev", [1, 2, 3, 8])
@pytest.mark.parametrize("shape,dtype", [((96, 280, 24), np.uint32), ((130, 97, 41), np.uint8), ((512, 96, 20), np.uint16),
                                         ((40, 1000, 9), np.uint64), ((1536, 100, 8), np.uint8)])
def test_virtual_devices_match_the_oracle(edt_gpu, oracle_port, ndev, shape, dtype):
    lab = voronoi_labels(shape, nseeds=40, seed=sum(shape), upsample=4, membrane=0.04).astype(dtype)
    for an, bb, sqrt in (((6.0, 6.0, 30.0), True, False), ((1.0, 1.5, 0.5), False, True)):
        want = oracle_port.edtsq(lab, an, bb)
        if sqrt:
            want = np.sqrt(want)
        from edt import _lib
        code = {1: _lib.U8, 2: _lib.U16, 4: _lib.U32, 8: _lib.U64}[np.dtype(dtype).itemsize]
        if not _lib.load().edt_hip_multi_supported(code, shape[0], shape[1], shape[2], ndev):
            # a volume that cannot be cut ndev ways is an error, not a silent single-device run
            with pytest.raises(_lib.EdtHipError) as ei:
                _multi(lab, an, bb, sqrt, [0] * ndev)
            assert ei.value.code == -4
            continue
        got = _multi(lab, an, bb, sqrt, [0] * ndev)
        assert np.array_equal(got, want, equal_nan=True), (ndev, shape, an, bb)
        # the pool is warm now: a second transform allocates nothing and gives the same values; chunk counts 1 and 3
        for chunks in ("1", "3"):
            os.environ["EDT_HIP_MULTI_CHUNKS"] = chunks
            try:
                assert np.array_equal(_multi(lab, an, bb, sqrt, [0] * ndev), want, equal_nan=True), (ndev, shape, chunks)
            finally:
                os.environ.pop("EDT_HIP_MULTI_CHUNKS", None)


def test_front_ends_take_the_route_when_devices_are_set(edt_gpu, oracle_port):
    """edt_hip_set_devices: the ordinary host-buffer entry points (and with them edt.edtsq / edt::edt<T>()) shard."""
    lab = voronoi_labels((160, 144, 96), nseeds=50, seed=5, upsample=4)
    want = oracle_port.edtsq(lab, (1.0, 1.0, 2.0), False)
    edt_gpu.set_devices([0, 0, 0, 0])
    try:
        assert np.array_equal(edt_gpu.edtsq(lab, anisotropy=(1.0, 1.0, 2.0)), want)
        assert np.array_equal(edt_gpu.edt(np.ascontiguousarray(lab), anisotropy=(1.0, 1.0, 2.0)), np.sqrt(want))
        # volumes the slab-record form does not cover fall back to one device
        small = lab[:, :40, :3]
        assert np.array_equal(edt_gpu.edtsq(small, anisotropy=(1.0, 1.0, 2.0)),
                              oracle_port.edtsq(small, (1.0, 1.0, 2.0), False))
    finally:
        edt_gpu.set_devices(None)
    with pytest.raises(Exception):
        edt_gpu.set_devices([0, 99])
