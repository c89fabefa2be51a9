"""GPU tier: pass X with four consecutive voxels per lane (csrc/edt_rowquad.hip: 16-byte loads, the left neighbour by a DPP
shift, run starts as four ballot masks + a nibble per lane) -- uint32 labels in the index form, rows of whole 16-byte granules
up to 1024 voxels.  Every case against the oracle AND against the one-voxel-per-lane kernel (debug bit 0x40000000), whose
outputs (16-bit indices and the three bit planes) it must reproduce bit for bit: the column passes behind both are the same."""
import numpy as np
import pytest

from synth import blocky_labels, voronoi_labels

pytestmark = pytest.mark.gpu


def both_kernels(edt_gpu, lab, an, bb, sqrt=False):
    from edt import _lib
    lib = _lib.load()
    fn = edt_gpu.edt if sqrt else edt_gpu.edtsq
    try:
        lib.edt_hip_set_debug_mode(0)
        new = fn(lab, anisotropy=an, black_border=bb)
        lib.edt_hip_set_debug_mode(0x40000000)
        old = fn(lab, anisotropy=an, black_border=bb)
    finally:
        lib.edt_hip_set_debug_mode(0)
    return new, old


@pytest.mark.parametrize("sx", [4, 8, 36, 60, 64, 252, 256, 260, 300, 508, 512, 516, 700, 768, 772, 1000, 1020, 1024])
def test_rowquad_row_lengths(edt_gpu, oracle_port, sx):
    rng = np.random.default_rng(sx)
    for shape in ((sx, 70, 37), (sx, 33, 1), (sx, 128, 2)):
        labs = [blocky_labels(shape, nlabels=5, zero_frac=0.2, block=int(rng.integers(1, 9)), rng=rng).astype(np.uint32),
                blocky_labels(shape, nlabels=3, zero_frac=0.0, block=int(rng.integers(30, 400)), rng=rng).astype(np.uint32),
                rng.integers(0, 3, size=shape).astype(np.uint32),
                np.ones(shape, dtype=np.uint32)]
        for lab in labs:
            lab = np.asfortranarray(lab)
            for an, bb in (((1, 1, 1), True), ((6, 6, 30), False), ((0.5, 2.0, 1.0), True)):
                want = oracle_port.edtsq(lab, an, bb)
                new, old = both_kernels(edt_gpu, lab, an, bb)
                assert np.array_equal(new, want, equal_nan=True), (shape, an, bb, "quad kernel vs oracle")
                assert np.array_equal(old, want, equal_nan=True), (shape, an, bb, "wave kernel vs oracle")


@pytest.mark.parametrize("shape", [(512, 200, 96), (256, 130, 130), (640, 97, 140), (1024, 64, 40), (132, 300, 100)])
def test_rowquad_segmentations(edt_gpu, oracle_port, shape):
    for lab in (voronoi_labels(shape, nseeds=40, seed=sum(shape), upsample=4, membrane=0.05),
                voronoi_labels(shape, nseeds=400, seed=1 + sum(shape), upsample=1)):
        lab = np.asfortranarray(lab.astype(np.uint32))
        for an, bb in (((1, 1, 1), False), ((6, 6, 30), True)):
            want = oracle_port.edtsq(lab, an, bb)
            new, old = both_kernels(edt_gpu, lab, an, bb)
            assert np.array_equal(new, want), (shape, an, bb)
            assert np.array_equal(old, want), (shape, an, bb)
        new, _ = both_kernels(edt_gpu, lab, (1, 1, 1), False, sqrt=True)
        assert np.array_equal(new, np.sqrt(oracle_port.edtsq(lab, (1, 1, 1), False)))


def test_rowquad_two_dimensional_and_c_order(edt_gpu, oracle_port):
    rng = np.random.default_rng(77)
    for shape in ((300, 260), (1024, 200), (40, 1024)):
        img = blocky_labels(shape, nlabels=6, zero_frac=0.1, block=13, rng=rng).astype(np.uint32)
        for order in ("F", "C"):
            a = np.asfortranarray(img) if order == "F" else np.ascontiguousarray(img)
            for an, bb in (((1, 1), True), ((2, 3), False)):
                want = oracle_port.edtsq(a, an, bb)
                new, old = both_kernels(edt_gpu, a, an, bb)
                assert np.array_equal(new, want) and np.array_equal(old, want), (shape, order, an, bb)
