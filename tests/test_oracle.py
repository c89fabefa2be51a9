"""CPU tests that PIN the oracle (oracle/edt_oracle.c) before anything is compared to it:
against the golden vectors recorded from the reference, the reference's hand-derivable known
answers, the brute-force specification, and -- where available -- the compiled reference."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import spec
from synth import blocky_labels, box_edtsq_closed_form

INF = np.inf
ALL_TYPES = [np.uint8, np.uint16, np.uint32, np.uint64, np.float32, bool]


def same(a, b):
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


# ---- golden vectors recorded from the real reference -----------------------------------
def test_golden_random(oracle_port):
    for c in load_golden("edt_random.npz"):
        lab = c["labels"]
        lab = np.asfortranarray(lab) if str(c["order"]) == "F" else np.ascontiguousarray(lab)
        an = tuple(c["anisotropy"])
        an = an[0] if lab.ndim == 1 else an
        bb = bool(c["black_border"])
        assert same(oracle_port.edtsq(lab, an, bb), c["edtsq"])
        assert same(oracle_port.edt(lab, an, bb), c["edt"])


def test_golden_configs(oracle_port):
    for c in load_golden("edt_configs.npz"):
        lab = np.asfortranarray(c["labels"])
        got = oracle_port.edtsq(lab, tuple(c["anisotropy"]), bool(c["black_border"]))
        assert same(got, c["edtsq"])


def test_golden_sdf_voxel_graph(oracle_port):
    for c in load_golden("edt_sdf_voxel_graph.npz"):
        lab, an, bb = c["labels"], tuple(c["anisotropy"]), bool(c["black_border"])
        if str(c["kind"]) == "sdf":
            got = oracle_port.sdf(lab, an, bb)
        else:
            got = oracle_port.edtsq(lab, an, bb, voxel_graph=c["graph"])
        assert same(got, c["out"])


# ---- known answers of the reference's own test-suite (automated_test.py) -----------------
@pytest.mark.parametrize("dtype", ALL_TYPES)
def test_one_d_known_answers(oracle_port, dtype):
    # automated_test.py:62-146
    def check(labels, ans, bb, anisotropy=1.0):
        got = oracle_port.edtsq(np.array(labels, dtype=dtype), anisotropy, bb)
        assert same(got, np.array(ans, dtype=np.float32))

    check([1], [1], True)
    check([5 if dtype is not bool else 1], [1], True)
    check([0, 1, 1, 1, 0], [0, 1, 4, 1, 0], True)
    check([1, 1, 1, 1], [1, 4, 4, 1], True)
    check([1, 1, 1, 1], [4, 16, 16, 4], True, anisotropy=2.0)
    check([1], [INF], False)
    check([0, 1, 1, 1, 0], [0, 1, 4, 1, 0], False)
    check([1, 1, 1, 1], [INF, INF, INF, INF], False)
    check([0, 1, 1, 1], [0, 1, 4, 9], False)
    check([1, 1, 1, 0], [9, 4, 1, 0], False)
    if dtype is not bool:
        check([1, 1, 1, 1, 1, 0, 2, 2, 2, 2, 2, 1, 1, 1, 1, 3],
              [1, 4, 9, 4, 1, 0, 1, 4, 9, 4, 1, 1, 4, 4, 1, 1], True)
        check([1, 1, 1, 1, 1, 0, 2, 2, 2, 2, 2, 1, 1, 1, 1, 3],
              [25, 16, 9, 4, 1, 0, 1, 4, 9, 4, 1, 1, 4, 4, 1, 1], False)


def test_two_d_known_answers(oracle_port):
    # identity-like images, automated_test.py:188-230
    lab = np.ones((5, 5), dtype=np.uint32)
    got = oracle_port.edtsq(lab, (1, 1), True)
    want = np.array([[1, 1, 1, 1, 1], [1, 4, 4, 4, 1], [1, 4, 9, 4, 1], [1, 4, 4, 4, 1],
                     [1, 1, 1, 1, 1]], dtype=np.float32)
    assert same(got, want)
    assert np.all(np.isinf(oracle_port.edtsq(lab, (1, 1), False)))
    # label boundary through the middle: each half is its own object
    lab = np.array([[1, 1, 2, 2]] * 4, dtype=np.uint8)
    got = oracle_port.edtsq(lab, (1, 1), False)
    assert same(got, np.array([[4, 1, 1, 4]] * 4, dtype=np.float32))
    # a single background pixel in a field, anisotropy (5, 6) (C order: rows are y)
    lab = np.ones((3, 3), dtype=np.uint16)
    lab[1, 1] = 0
    got = oracle_port.edtsq(lab, (5, 6), False)
    want = np.array([[61, 25, 61], [36, 0, 36], [61, 25, 61]], dtype=np.float32)
    assert same(got, want)


def test_three_d_cube_known_answers(oracle_port):
    # automated_test.py:426-551 style: 3x3x3 cube, centre voxel distances
    lab = np.ones((3, 3, 3), dtype=np.uint32)
    got = oracle_port.edtsq(lab, (4, 4, 4), True)
    assert got[1, 1, 1] == 64 and got[0, 0, 0] == 16 and got[1, 1, 0] == 16
    got = oracle_port.edtsq(lab, (6, 6, 5), True)
    assert got[1, 1, 1] == 100.0  # 2 voxels * 5 along the cheapest axis
    assert same(got, np.ascontiguousarray(box_edtsq_closed_form((3, 3, 3), (6, 6, 5))))


def test_box_closed_form(oracle_port):
    for shape, an in (((17, 9, 23), (6, 6, 30)), ((8, 8, 8), (1, 1, 1)), ((5, 31, 2), (3, 1, 2))):
        lab = np.ones(shape, dtype=np.uint32, order="F")
        assert same(oracle_port.edtsq(lab, an, True), box_edtsq_closed_form(shape, an))


def test_scaling_identity(oracle_port):
    # automated_test.py:632-649: integer anisotropy scales the squared transform exactly
    rng = np.random.default_rng(3)
    lab = blocky_labels((20, 18, 16), nlabels=3, zero_frac=0.3, block=3, rng=rng).astype(np.uint8)
    base = oracle_port.edt(lab, (1, 1, 1), True)
    for w in (2, 7, 149):
        assert same(oracle_port.edt(lab, (w, w, w), True), (w * base).astype(np.float32))


def test_all_inf_and_empty(oracle_port):
    assert np.all(np.isinf(oracle_port.edt(np.ones((6, 5, 4), dtype=np.uint8), (1, 1, 1), False)))
    assert oracle_port.edtsq(np.zeros((0,), dtype=np.uint8)).shape == (0,)
    assert oracle_port.edtsq(np.zeros((3, 0, 2), dtype=np.uint8)).shape == (3, 0, 2)


def test_c_vs_f_order(oracle_port):
    rng = np.random.default_rng(4)
    lab = blocky_labels((13, 21, 8), nlabels=4, zero_frac=0.2, block=3, rng=rng).astype(np.uint32)
    a = oracle_port.edtsq(np.ascontiguousarray(lab), (2, 3, 5), False)
    b = oracle_port.edtsq(np.asfortranarray(lab), (2, 3, 5), False)
    assert same(a, b)


# ---- brute-force specification -------------------------------------------------------------
def test_against_bruteforce_spec(oracle_port):
    rng = np.random.default_rng(11)
    for t in range(40):
        dims = int(rng.integers(1, 4))
        shape = tuple(int(rng.integers(1, 13)) for _ in range(dims))
        lab = blocky_labels(shape, nlabels=3, zero_frac=0.3, block=int(rng.integers(1, 4)), rng=rng)
        lab = np.asfortranarray(lab.astype(np.uint16))
        an = [(1, 1, 1), (6, 6, 30), (0.5, 0.7, 1.3), (2, 1, 3)][t % 4][:dims]
        bb = bool(t % 2)
        want = spec.edtsq_xfast(lab, an, bb)
        got = oracle_port.edtsq(lab, an[0] if dims == 1 else an, bb)
        assert same(got, want), (t, shape, an, bb)


# ---- the compiled reference itself -----------------------------------------------------------
def test_against_compiled_reference(oracle_port, oracle_ref):
    rng = np.random.default_rng(12)
    dtypes = [np.uint8, np.uint16, np.uint32, np.uint64, np.float32, np.float64, bool, np.int16]
    for t in range(120):
        dims = int(rng.integers(1, 4))
        shape = tuple(int(rng.integers(1, 48)) for _ in range(dims))
        lab = blocky_labels(shape, nlabels=int(rng.integers(1, 9)), zero_frac=float(rng.random() * 0.5),
                            block=int(rng.integers(1, 7)), rng=rng).astype(dtypes[t % len(dtypes)])
        if t % 3 == 0:
            lab = np.asfortranarray(lab)
        an = [(1, 1, 1), (6, 6, 30), (0.5, 0.7, 1.3), (4, 4, 40), (1e-3, 2.5, 7)][t % 5][:dims]
        an = an[0] if dims == 1 else an
        bb = bool(rng.integers(0, 2))
        assert same(oracle_port.edtsq(lab, an, bb), oracle_ref.edtsq(lab, an, bb, parallel=1 + t % 2))


def test_reference_fast_math_twin_agrees(oracle_ref):
    # the setup.py-flag build and the strict build agree bit-for-bit
    from oracle import harness
    fast = harness.ref(fast=True)
    rng = np.random.default_rng(13)
    lab = np.asfortranarray(blocky_labels((40, 36, 28), 12, 0.1, 5, rng).astype(np.uint32))
    for an, bb in (((6, 6, 30), True), ((0.5, 0.7, 1.3), False)):
        assert same(oracle_ref.edtsq(lab, an, bb), fast.edtsq(lab, an, bb))


def test_binary_route_against_compiled_reference(oracle_port, oracle_ref):
    # pyedt::_binary_edt{2,3}dsq<T> for multi-valued non-bool T (src/edt.hpp:487-576, :681-732): labels split runs in
    # pass 1 only; it differs from the multi-label transform as soon as two non-zero labels touch along y or z
    rng = np.random.default_rng(21)
    dtypes = [np.uint8, np.uint16, np.uint32, np.uint64, np.float32, np.float64, bool]
    differs = 0
    for t in range(84):
        dims = 2 + t % 2
        shape = tuple(int(rng.integers(1, 40)) for _ in range(dims))
        lab = blocky_labels(shape, nlabels=int(rng.integers(1, 6)), zero_frac=float(rng.random() * 0.5),
                            block=int(rng.integers(1, 7)), rng=rng).astype(dtypes[t % len(dtypes)])
        if t % 3 == 0:
            lab = np.asfortranarray(lab)
        an = [(1, 1, 1), (6, 6, 30), (0.5, 0.7, 1.3), (4, 4, 40)][t % 4][:dims]
        bb = bool(rng.integers(0, 2))
        want = oracle_ref.binary_edtsq(lab, an, bb, parallel=1 + t % 2)
        assert same(oracle_port.binary_edtsq(lab, an, bb), want), (t, shape, an, bb)
        differs += not same(want, oracle_ref.edtsq(lab, an, bb))
    assert differs > 10  # the case the facade used to get wrong is really exercised
