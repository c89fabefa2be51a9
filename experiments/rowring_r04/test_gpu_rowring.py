"""Pass X, index form, rows staged in an LDS ring by LDS-DMA (csrc/edt_rowring.hip: rows of exactly 512 / 1024 voxels,
uint8 / bool / uint32 labels): against the oracle and against the register-pipelined kernel (debug bit 0x400000) --
short and partial y-bands, one slice, many slices (the ring runs across group boundaries), 2-D, both border modes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _labels(shape, dtype, rng, p_bg=0.1):
    from synth import blocky_labels
    lab = blocky_labels(shape, nlabels=7, zero_frac=p_bg, block=9, rng=rng)
    if dtype == np.bool_:
        return np.asfortranarray(lab != 0)
    return np.asfortranarray(lab.astype(dtype))


@pytest.mark.parametrize("shape,dtype", [
    ((512, 37, 5), np.uint32), ((512, 64, 3), np.uint8), ((512, 1, 1), np.uint32), ((512, 33, 40), np.uint32),
    ((1024, 33, 2), np.uint32), ((1024, 70, 3), np.uint8), ((512, 40), np.uint32), ((1024, 31), np.bool_),
    ((512, 96, 17), np.bool_),
])
def test_rowring_against_oracle_and_register_kernel(edt_gpu, oracle_port, shape, dtype):
    from edt import _lib
    lib = _lib.load()
    rng = np.random.default_rng(sum(shape))
    lab = _labels(shape, dtype, rng)
    for an, bb in (((1.0, 1.0, 1.0), True), ((6.0, 6.0, 30.0), False), ((2.0, 1.0, 3.0), True)):
        an = an[:len(shape)]
        want = oracle_port.edtsq(lab, an, bb)
        try:
            for mode in (0, 0x400000):
                lib.edt_hip_set_debug_mode(mode)
                got = edt_gpu.edtsq(lab, anisotropy=an, black_border=bb)
                assert np.array_equal(got, want, equal_nan=True), (shape, dtype, an, bb, hex(mode))
        finally:
            lib.edt_hip_set_debug_mode(0)


def test_rowring_rows_without_any_boundary(edt_gpu, oracle_port):
    """one label everywhere, no black border: every index is 0xFFFF (no boundary on either side)"""
    lab = np.ones((512, 40, 3), dtype=np.uint32, order="F")
    lab[100:140, 7, 1] = 0
    want = oracle_port.edtsq(lab, (1.0, 1.0, 1.0), False)
    got = edt_gpu.edtsq(lab, anisotropy=(1.0, 1.0, 1.0), black_border=False)
    assert np.array_equal(got, want, equal_nan=True)
