"""GPU tier: the entry points around the 3-D hot path: the parallel 1-D pipeline for lines
of any length, stacks of 2-D images in one batch, device-side run extraction / per-label images, sdf in one
round trip, the concurrent first touch of the host result buffer."""
import numpy as np
import pytest

from synth import blocky_labels, voronoi_labels

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 2, 63, 64, 1023, 1024, 1025, 5000, 70001, 1 << 20])
def test_line_of_any_length(edt_gpu, oracle_port, n):
    rng = np.random.default_rng(n)
    for kind in ("ones", "blocky", "noise"):
        if kind == "ones":
            lab = np.ones(n, dtype=np.uint32)
        elif kind == "blocky":
            lab = blocky_labels((n,), nlabels=5, zero_frac=0.2, block=int(rng.integers(1, 4000)), rng=rng).astype(np.uint16)
        else:
            lab = rng.integers(0, 3, size=n).astype(np.uint8)
        for w in (1.0, 6.0, 0.7, 1e-3):          # exact multiples and a tabulated sequential sum
            for bb in (True, False):
                want = oracle_port.edtsq(lab, w, bb)
                got = edt_gpu.edtsq(lab, anisotropy=w, black_border=bb)
                assert np.array_equal(got, want), (n, kind, w, bb)
                assert np.array_equal(edt_gpu.edt(lab, anisotropy=w, black_border=bb), np.sqrt(want))


def test_line_device_resident_and_generic_agree(edt_gpu, oracle_port):
    import torch
    from edt import device
    rng = np.random.default_rng(3)
    lab = blocky_labels((300000,), nlabels=7, zero_frac=0.1, block=977, rng=rng).astype(np.int32)
    want = oracle_port.edtsq(lab, 2.5, False)
    t = torch.from_numpy(lab).cuda()
    assert np.array_equal(device.edtsq(t, anisotropy=2.5).cpu().numpy(), want)
    plan = device.Plan((lab.size,), 2, t.device)
    got = plan.run(t, (2.5,), black_border=False, force_generic=True)   # the one-thread port of the reference loop
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("shape", [(1, 5, 7), (9, 64, 64), (33, 130, 96), (300, 40, 52), (4, 513, 1000)])
def test_stack_of_images_equals_image_by_image(edt_gpu, oracle_port, shape):
    import torch
    from edt import device
    rng = np.random.default_rng(sum(shape))
    stack = blocky_labels(shape, nlabels=6, zero_frac=0.15, block=7, rng=rng).astype(np.uint32)
    for an, bb in (((1.0, 1.0), False), ((3.0, 0.5), True)):
        want = np.stack([oracle_port.edtsq(img, an, bb) for img in stack])
        got = edt_gpu.edtsq_stack(stack, anisotropy=an, black_border=bb)
        assert got.shape == want.shape and np.array_equal(got, want), (shape, an, bb)
        assert np.array_equal(edt_gpu.edt_stack(stack, anisotropy=an, black_border=bb), np.sqrt(want))
        t = torch.from_numpy(stack.view(np.int32)).cuda()
        assert np.array_equal(device.edtsq_stack(t, anisotropy=an, black_border=bb).cpu().numpy(), want)


@pytest.mark.parametrize("dtype", ["uint8", "int32", "int64", "float32"])
def test_device_runs_equal_host_runs(edt_gpu, dtype):
    import torch
    from edt import device
    rng = np.random.default_rng(12)
    for shape in ((1,), (5000,), (37, 41), (20, 33, 29)):
        lab = blocky_labels(shape, nlabels=6, zero_frac=0.2, block=3, rng=rng).astype(dtype)
        starts, ends, values = device.runs(torch.from_numpy(lab).cuda())
        host = edt_gpu.runs(lab)                       # {label: [(start, end), ...]}
        flat = sorted((s, e, k) for k, rr in host.items() for s, e in rr)
        assert starts.tolist() == [s for s, _, _ in flat]
        assert ends.tolist() == [e for _, e, _ in flat]
        assert values.cpu().numpy().tolist() == [k for _, _, k in flat]
    s, e, v = device.runs(torch.zeros(0, dtype=torch.int32, device="cuda"))
    assert s.numel() == 0 and e.numel() == 0


def test_device_each_large_labels_in_place(edt_gpu):
    import torch
    from edt import device
    lab = voronoi_labels((96, 80, 72), nseeds=40, seed=2, upsample=4, membrane=0.05)
    tl = torch.from_numpy(np.ascontiguousarray(lab.T).view(np.int32)).cuda()
    dt = device.edt(tl, anisotropy=(3, 2, 1))
    mdt, mlab = dt.cpu().numpy(), tl.cpu().numpy()
    for in_place in (False, True):
        it = device.each(tl, dt, in_place=in_place)
        assert len(it) == len(np.unique(mlab)) - int(0 in mlab)
        for k, img in it:
            assert np.array_equal(img.cpu().numpy(), (mlab == k) * mdt), (k, in_place)


def test_negative_voxel_size_along_y_or_z_is_its_magnitude(edt_gpu, oracle_port):
    """along y and z a voxel size enters the reference only as its square (src/edt.hpp:181, :258), so a negative
    wy / wz gives the field of |w| there -- and here (the C ABI drops the sign; only wx must be positive)."""
    rng = np.random.default_rng(78)
    lab = np.asfortranarray(blocky_labels((44, 130, 100), nlabels=5, zero_frac=0.2, block=9, rng=rng).astype(np.uint32))
    for an in ((2.0, -3.0, 5.0), (1.0, 1.0, -1.0), (6.0, -6.0, -30.0), (0.7, -1.3, 2.0)):
        for bb in (True, False):
            want = oracle_port.edtsq(lab, an, bb)   # the reference's arithmetic with the sign as given
            assert np.array_equal(want, oracle_port.edtsq(lab, tuple(abs(a) for a in an), bb))
            assert np.array_equal(edt_gpu.edtsq(lab, anisotropy=an, black_border=bb), want), (an, bb)
    with pytest.raises(ValueError):
        edt_gpu.edtsq(lab, anisotropy=(-2.0, 3.0, 5.0))


def test_signed_transform_is_the_two_transform_definition(edt_gpu, oracle_port):
    """sdf / sdfsq as ONE transform (EDT_FLAG_SIGNED, round 6): label 0 measured like every other label, its voxels negated --
    against the definition edt(x) - edt(x == 0) computed by the oracle, and against this library's own two-transform
    composition, over label types, orders, border modes, voxel sizes with and without a quantum, 2-D and 3-D."""
    import torch
    from edt import _lib, device

    rng = np.random.default_rng(79)
    cases = [((48, 140, 132), (6.0, 6.0, 30.0)), ((40, 100, 36), (1.0, 1.0, 1.0)), ((33, 70, 41), (0.7, 1.3, 2.0)),
             ((64, 200), (1.0, 2.0)), ((130, 37), (3.58, 4.0)), ((512, 40, 40), (4.0, 4.0, 40.0))]
    dtypes = [np.uint8, np.uint16, np.uint32, np.uint64, np.float32, bool]
    for i, (shape, an) in enumerate(cases):
        lab = blocky_labels(shape, nlabels=4, zero_frac=0.45, block=int(rng.integers(3, 12)), rng=rng).astype(dtypes[i % len(dtypes)])
        code = {1: _lib.U8, 2: _lib.U16, 4: _lib.U32, 8: _lib.U64}[lab.dtype.itemsize] if lab.dtype.kind in "ub" else _lib.F32
        ext = tuple(shape[::-1]) + (1,) * (3 - len(shape))   # C order: x is the last axis
        assert _lib.load().edt_hip_signed_supported(code, len(shape), *ext, 0) == 1
        for bb in (True, False):
            for arr in (lab, np.asfortranarray(lab)):
                want, wantsq = oracle_port.sdf(arr, an, bb), oracle_port.sdfsq(arr, an, bb)
                assert (want < 0).any() and (want > 0).any()
                assert np.array_equal(edt_gpu.sdf(arr, anisotropy=an, black_border=bb), want, equal_nan=True), (shape, an, bb)
                assert np.array_equal(edt_gpu.sdfsq(arr, anisotropy=an, black_border=bb), wantsq, equal_nan=True), (shape, an, bb)
            t = torch.from_numpy(np.ascontiguousarray(lab)).cuda()
            one = device.sdf(t, anisotropy=an, black_border=bb)
            two = device._signed(t, an, bb, sqrt=True, one_transform=False)
            assert torch.equal(one, two) and np.array_equal(one.cpu().numpy(), oracle_port.sdf(lab, an, bb), equal_nan=True)
            # (the sign as the epilogue of the last integer pass where the host can prove both passes stay on that kernel -- the
            # default -- and as a pass of its own: debug bit 0x400)
            try:
                _lib.load().edt_hip_set_debug_mode(0x400)
                assert torch.equal(device.sdf(t, anisotropy=an, black_border=bb), one), (shape, an, bb, "sign as a pass of its own")
            finally:
                _lib.load().edt_hip_set_debug_mode(0)
            assert torch.equal(device.sdfsq(t, anisotropy=an, black_border=bb), device._signed(t, an, bb, sqrt=False, one_transform=False))
    # a volume without any background, without a black border: +inf everywhere, as the definition says (inf - 0)
    ones = np.ones((36, 100, 100), dtype=np.uint8)
    assert np.array_equal(edt_gpu.sdf(ones, black_border=False), oracle_port.sdf(ones, None, False))
    # all background: -inf
    assert np.array_equal(edt_gpu.sdf(np.zeros_like(ones), black_border=False), oracle_port.sdf(np.zeros_like(ones), None, False))


def test_sdf_single_round_trip(edt_gpu, oracle_port):
    rng = np.random.default_rng(77)
    for shape in ((200,), (70, 45), (40, 52, 36)):
        lab = blocky_labels(shape, nlabels=3, zero_frac=0.4, block=5, rng=rng).astype(np.uint16)
        an = (2.0, 1.0, 3.0)[:len(shape)]
        an = an[0] if len(shape) == 1 else an
        for bb in (True, False):
            for arr in (lab, np.asfortranarray(lab)):
                assert np.array_equal(edt_gpu.sdf(arr, anisotropy=an, black_border=bb), oracle_port.sdf(arr, an, bb),
                                      equal_nan=True)
                assert np.array_equal(edt_gpu.sdfsq(arr, anisotropy=an, black_border=bb),
                                      oracle_port.sdfsq(arr, an, bb), equal_nan=True)


def test_large_result_buffer_is_prefaulted_correctly(edt_gpu, oracle_port):
    """>= 32 MiB results are first-touched by helper threads while the labels travel (csrc/edt_host.hip:
    Prefault): the bytes that come back are still exactly the transform."""
    lab = voronoi_labels((320, 256, 128), nseeds=300, seed=9, upsample=4)     # 40 MiB result
    want = oracle_port.edtsq(lab, (1.0, 1.0, 2.0), False)
    for _ in range(2):
        assert np.array_equal(edt_gpu.edtsq(lab, anisotropy=(1.0, 1.0, 2.0), black_border=False), want)


def test_line_without_workspace_is_served(edt_gpu, oracle_port):
    """edt_hip_edtsq_device(ndim = 1, d_workspace = NULL): the round-1 ABI needed no scratch for a line; the parallel
    pipeline does, and a call without it is served by the row kernel instead of being refused."""
    import ctypes
    import torch
    from edt import _lib
    lib = _lib.load()
    rng = np.random.default_rng(4)
    lab = blocky_labels((20000,), nlabels=4, zero_frac=0.2, block=311, rng=rng).astype(np.uint16)
    t = torch.from_numpy(lab.view(np.int16)).cuda()
    out = torch.empty(lab.size, dtype=torch.float32, device="cuda")
    for bb in (False, True):
        rc = lib.edt_hip_edtsq_device(t.data_ptr(), _lib.U16, 1, lab.size, 1, 1, 3.0, 1.0, 1.0,
                                      _lib.FLAG_BLACK_BORDER if bb else 0, out.data_ptr(), None, 0, None)
        _lib.check(rc)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), oracle_port.edtsq(lab, 3.0, bb))
    # 2-D without a workspace is still an argument error (it has no scratch-free form)
    rc = lib.edt_hip_edtsq_device(t.data_ptr(), _lib.U16, 2, 100, 200, 1, 1.0, 1.0, 1.0, 0, out.data_ptr(), None, 0, None)
    assert rc == -2


def test_binary_route_through_every_kernel_family(edt_gpu, oracle_port):
    """EDT_FLAG_BINARY_YZ (one all-foreground run per column) through the other forms of the same passes: fp32 instead
    of 16-bit indices between X and Y, hulls only, windows on every tile, the workgroup-phased row / column kernels, and
    the size-agnostic kernels (long axis)."""
    from edt import _lib
    lib = _lib.load()
    rng = np.random.default_rng(8)
    lab = np.asfortranarray(blocky_labels((70, 90, 41), nlabels=3, zero_frac=0.35, block=6, rng=rng).astype(np.uint16))
    want = oracle_port.binary_edtsq(lab, (2.0, 1.0, 3.0), False)
    try:
        for mode in (0, 0x100000, 0x2000, 0x4000, 0xC000, 32 | 64):
            lib.edt_hip_set_debug_mode(mode)
            got = edt_gpu.binary_edtsq(lab, anisotropy=(2.0, 1.0, 3.0), black_border=False)
            assert np.array_equal(got, want), hex(mode)
    finally:
        lib.edt_hip_set_debug_mode(0)
    # device-resident form (C-ordered tensor: anisotropy in (z, y, x) order)
    import torch
    from edt import device
    t = torch.from_numpy(np.ascontiguousarray(lab.T).view(np.int16)).cuda()
    got = device.binary_edtsq(t, anisotropy=(3.0, 1.0, 2.0), black_border=False).cpu().numpy().T
    assert np.array_equal(got, want)
    long_axis = np.asfortranarray(blocky_labels((5, 33000), nlabels=3, zero_frac=0.3, block=700, rng=rng).astype(np.uint8))
    assert np.array_equal(edt_gpu.binary_edtsq(long_axis, anisotropy=(1.0, 2.0), black_border=True),
                          oracle_port.binary_edtsq(long_axis, (1.0, 2.0), True))
    long_row = np.asfortranarray(blocky_labels((3000, 7, 3), nlabels=3, zero_frac=0.3, block=300, rng=rng).astype(np.uint32))
    assert np.array_equal(edt_gpu.binary_edtsq(long_row, anisotropy=(1.0, 2.0, 0.5), black_border=False),
                          oracle_port.binary_edtsq(long_row, (1.0, 2.0, 0.5), False))


def test_very_long_line_with_tabulated_sums(edt_gpu, oracle_port):
    """A 2^24-voxel run at voxel sizes whose multiples are NOT exact: the table of sequential fp32 sums is built in
    parallel (every 1024-entry chunk starts from a value reached by jumping through the binades, csrc/edt_seqsum.h) and
    must equal the reference's running sums far beyond 2^23, where 0.7 has long since started to round to 1.0 per step."""
    n = 1 << 24
    lab = np.ones(n, dtype=np.uint8)
    lab[n // 3] = 0                     # two runs of ~5.6 and ~11.2 million voxels
    for w in (0.7, 0.1):
        for bb in (True, False):
            want = oracle_port.edtsq(lab, w, bb)
            got = edt_gpu.edtsq(lab, anisotropy=w, black_border=bb)
            assert np.array_equal(got, want), (w, bb, np.argwhere(got != want)[:3])


_CELLS = {}


def _cells(nseeds, shape):
    if (nseeds, shape) not in _CELLS:
        _CELLS[(nseeds, shape)] = voronoi_labels(shape, nseeds, seed=nseeds, upsample=1 if nseeds < 100 else 2,
                                                 membrane=0.0 if nseeds < 100 else 0.01)
    return _CELLS[(nseeds, shape)]


@pytest.mark.parametrize("anisotropy", [(3.58, 3.58, 40.0), (1.1, 1.1, 1.1), (0.1, 0.3, 0.2), (0.7, 1.3, 2.1), (30.0, 6.0, 6.0),
                                        (1.0e-3, 1.0, 1.0), (7.25, 0.5, 1.3)])
def test_inexact_voxel_sizes_both_candidate_forms(edt_gpu, oracle_port, anisotropy):
    """Voxel sizes whose c_d = w2 * d^2 are not exactly representable in fp32: the windowed path forms its candidates as
    fp32 fma's where the reference's fp64 sums are exact (edt_colwave_lane.h: brute_f32e_prefix) and as fp64 sums
    otherwise (debug bit 0x2000000 keeps the latter everywhere) -- both against the oracle, on cells of ~20 and ~45 voxels
    (windows of a few and of tens of rows) with and without the border."""
    from edt import _lib
    lib = _lib.load()
    for nseeds, shape in ((300, (160, 144, 130)), (24, (130, 160, 150))):
        lab = _cells(nseeds, shape)
        for bb in (False, True):
            want = oracle_port.edtsq(lab, anisotropy, bb)
            try:
                for mode in (0, 0x2000000, 0x100000):
                    lib.edt_hip_set_debug_mode(mode)
                    got = edt_gpu.edtsq(lab, anisotropy=anisotropy, black_border=bb)
                    assert np.array_equal(got, want), (anisotropy, bb, hex(mode), int((got != want).sum()))
            finally:
                lib.edt_hip_set_debug_mode(0)


class _CudaArrayInterfaceOnly:
    """a device array that speaks ONLY the CUDA array interface (what a Numba device array or a CuPy array offers)"""

    def __init__(self, t, typestr):
        self._keep = t
        self.__cuda_array_interface__ = {"shape": tuple(t.shape), "typestr": typestr, "data": (t.data_ptr(), False),
                                         "version": 3, "strides": None}


class _DLPackOnly:
    """a device array that speaks ONLY DLPack"""

    def __init__(self, t):
        self._t = t

    def __dlpack__(self, *a, **k):
        return self._t.__dlpack__(*a, **k)

    def __dlpack_device__(self):
        return self._t.__dlpack_device__()


def test_device_entry_points_take_dlpack_and_cuda_array_interface(edt_gpu, oracle_port):
    """callers whose arrays are not torch tensors (CuPy, Numba, JAX) reach the device entry points zero-copy
    through __dlpack__ / __cuda_array_interface__ (boundary: src/edt.pyx:639-734)."""
    import torch
    from edt import device
    lab = voronoi_labels((96, 80, 72), nseeds=40, seed=3, upsample=4, membrane=0.03)
    want = oracle_port.edtsq(lab, (6.0, 6.0, 30.0), True)
    t = torch.from_numpy(np.ascontiguousarray(lab.T).view(np.int32)).cuda()       # (z, y, x), x fastest
    ref = device.edtsq(t, anisotropy=(30.0, 6.0, 6.0), black_border=True)
    assert np.array_equal(ref.cpu().numpy().T, want)
    for wrapped in (_DLPackOnly(t), _CudaArrayInterfaceOnly(t, "<i4"), _CudaArrayInterfaceOnly(t.view(torch.uint8).view(torch.int32), "<u4")):
        got = device.edtsq(wrapped, anisotropy=(30.0, 6.0, 6.0), black_border=True)
        assert torch.equal(got, ref), type(wrapped).__name__
    assert torch.equal(device.as_device_tensor(_DLPackOnly(t)), t)
    assert device.as_device_tensor(_CudaArrayInterfaceOnly(t, "<i4")).data_ptr() == t.data_ptr()      # zero-copy
    assert torch.equal(device.sdf(_DLPackOnly(t), anisotropy=(30.0, 6.0, 6.0)), device.sdf(t, anisotropy=(30.0, 6.0, 6.0)))
    # the result speaks both protocols in the other direction
    assert hasattr(ref, "__dlpack__") and ref.__cuda_array_interface__["data"][0] == ref.data_ptr()
    with pytest.raises(TypeError):
        device.edtsq(np.zeros((4, 4), np.uint8))      # a host array: not this module's business
    with pytest.raises(TypeError):
        device.edtsq([[1, 2], [3, 4]])
