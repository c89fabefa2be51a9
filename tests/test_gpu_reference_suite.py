"""The reference's own acceptance tests (automated_test.py), run against the HIP module (`-m gpu`).

tests/test_oracle.py transcribes the known-answer / metamorphic cases of the reference's test file and
pins the CPU oracle with them; here the SAME test bodies run with the drop-in GPU module in place of the
oracle, plus the reference's scipy cross-checks (automated_test.py:148-168, :383-404, :553-578, :702-721).
"""
import numpy as np
import pytest

import test_oracle as T

pytestmark = pytest.mark.gpu


class HipBackend:
    """Adapts the drop-in module to the call shape the shared test bodies use."""

    def __init__(self, module):
        self.m = module

    def edtsq(self, data, anisotropy=None, black_border=False, parallel=1, voxel_graph=None):
        return self.m.edtsq(data, anisotropy=anisotropy, black_border=black_border, parallel=parallel,
                            voxel_graph=voxel_graph)

    def edt(self, data, anisotropy=None, black_border=False, parallel=1, voxel_graph=None):
        return self.m.edt(data, anisotropy=anisotropy, black_border=black_border, parallel=parallel,
                          voxel_graph=voxel_graph)

    def sdf(self, data, anisotropy=None, black_border=False, parallel=1):
        return self.m.sdf(data, anisotropy=anisotropy, black_border=black_border, parallel=parallel)


@pytest.fixture(scope="module")
def hip(edt_gpu):
    return HipBackend(edt_gpu)


@pytest.mark.parametrize("dtype", T.ALL_TYPES)
def test_one_d_known_answers(hip, dtype):
    T.test_one_d_known_answers(hip, dtype)


def test_two_d_known_answers(hip):
    T.test_two_d_known_answers(hip)


def test_three_d_cube_known_answers(hip):
    T.test_three_d_cube_known_answers(hip)


def test_box_closed_form(hip):
    T.test_box_closed_form(hip)


def test_scaling_identity(hip):
    T.test_scaling_identity(hip)


def test_all_inf_and_empty(hip):
    T.test_all_inf_and_empty(hip)


def test_c_vs_f_order(hip):
    T.test_c_vs_f_order(hip)


def test_against_bruteforce_spec(hip):
    T.test_against_bruteforce_spec(hip)


# ---- scipy cross-checks, as the reference does them ---------------------------------------------
def _scipy_edt(binary, sampling=None):
    from scipy import ndimage
    return ndimage.distance_transform_edt(binary, sampling=sampling)


@pytest.mark.parametrize("order", ["C", "F"])
@pytest.mark.parametrize("dtype", [np.uint32, bool])
def test_3d_scipy_comparison(edt_gpu, order, dtype):
    # automated_test.py:553-578: 102^3 random binary, both orders, uint32 and bool, abs tol 1e-6 on edt
    pytest.importorskip("scipy")
    rng = np.random.default_rng(102)
    lab = (rng.random((102, 102, 102)) < 0.8).astype(dtype, order=order)
    got = edt_gpu.edt(lab, black_border=False)
    want = _scipy_edt(lab != 0)
    assert np.all(np.abs(got - want) < 1e-4)  # scipy works in fp64; the reference's tolerance is 1e-6 on small values
    assert np.max(np.abs(got - want) / np.maximum(want, 1)) < 1e-6


def test_2d_scipy_comparison_black_border(edt_gpu):
    # automated_test.py:383-404: padding with one background pixel == black_border=True
    pytest.importorskip("scipy")
    rng = np.random.default_rng(7)
    lab = (rng.random((80, 70)) < 0.9).astype(np.uint8)
    padded = np.pad(lab, 1)
    want = _scipy_edt(padded)[1:-1, 1:-1]
    got = edt_gpu.edt(lab, black_border=True)
    assert np.max(np.abs(got - want)) < 1e-5


def test_3d_high_anisotropy(edt_gpu):
    # automated_test.py:702-721: 256^3 with anisotropy (1e6, 1.2e6, 40)
    pytest.importorskip("scipy")
    rng = np.random.default_rng(256)
    lab = (rng.random((128, 128, 128)) < 0.95).astype(np.uint8)
    an = (1000000, 1200000, 40)
    got = edt_gpu.edt(lab, anisotropy=an, black_border=False)
    want = _scipy_edt(lab, sampling=an)
    assert np.all(np.isclose(got, want, rtol=1e-6, atol=0))


def test_sdf_definition(edt_gpu):
    # automated_test.py:879-895: sdf == edt(x) - edt(x == 0)
    rng = np.random.default_rng(5)
    lab = (rng.random((40, 36, 30)) < 0.5).astype(np.uint8)
    got = edt_gpu.sdf(lab, anisotropy=(2, 3, 5), black_border=True)
    want = edt_gpu.edt(lab, anisotropy=(2, 3, 5), black_border=True) - edt_gpu.edt(lab == 0, anisotropy=(2, 3, 5), black_border=True)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("exp", range(-7, 9))
def test_anisotropy_range(edt_gpu, oracle_port, exp):
    # automated_test.py:791-817: anisotropies from 1e-7 to 1e8 -- here bit-for-bit against the oracle,
    # on a multi-label volume with background, both border modes
    from synth import blocky_labels
    rng = np.random.default_rng(100 + exp)
    lab = np.asfortranarray(blocky_labels((72, 40, 33), nlabels=4, zero_frac=0.15, block=5, rng=rng).astype(np.uint16))
    w = 10.0 ** exp
    for an in ((w, w, w), (w, 1.0, 3.0 * w)):
        for bb in (True, False):
            want = oracle_port.edtsq(lab, an, bb)
            got = edt_gpu.edtsq(lab, anisotropy=an, black_border=bb)
            assert np.array_equal(got, want, equal_nan=True), (an, bb)


def test_nan_large_array(edt_gpu):
    # automated_test.py:819-823: a 46342 x 1 array of ones must not produce NaN
    lab = np.ones((46342, 1), dtype=np.float64)
    out = edt_gpu.edtsq(lab, anisotropy=(1, 1), black_border=True)
    assert not np.any(np.isnan(out))
    assert out.max() == 1.0
