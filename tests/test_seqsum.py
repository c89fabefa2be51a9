"""CPU test of csrc/edt_seqsum.h: T[k] = the k-fold SEQUENTIAL fp32 sum of the voxel size (the reference's pass 1,
src/edt.hpp:92-114) by jumping through binades, against the plain loop."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "_build")


@pytest.fixture(scope="module")
def lib():
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, f"libseqsum_{os.getpid()}.so")
    subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC",
                    "-I" + os.path.join(ROOT, "euclidean-distance-transform-3d_amd", "csrc"),
                    os.path.join(ROOT, "tests", "seqsum_shim.cpp"), "-o", so], check=True)
    lib = ctypes.CDLL(so)
    lib.seqsum_check.restype = ctypes.c_longlong
    lib.seqsum_check.argtypes = [ctypes.c_float, ctypes.c_void_p, ctypes.c_longlong]
    lib.seqsum_jump.restype = ctypes.c_float
    lib.seqsum_jump.argtypes = [ctypes.c_float, ctypes.c_longlong]
    yield lib
    try:
        os.remove(so)
    except OSError:
        pass


def _check(lib, w, ks):
    ks = np.unique(np.asarray(ks, dtype=np.int64))
    return lib.seqsum_check(ctypes.c_float(w), ks.ctypes.data_as(ctypes.c_void_p), len(ks))


def test_jump_equals_loop(lib):
    rng = np.random.default_rng(1)
    ws = [0.1, 0.7, 1.3, 7.25, 3.0, 1e-3, 1e-7, 1.5, 0.75, 2.0 ** -20 * 3, 1.0, 6.0, 30.0, 0.3, 1.1, 16381.0, 1e6,
          float(np.float32(1) + np.float32(2.0 ** -23)), 2.0 ** -149 * 5, 1e-38, 3.3e38]
    ws += [float(np.float32(rng.uniform(0.01, 50.0))) for _ in range(60)]
    ws += [float(np.float32(rng.integers(1, 2 ** 24) * 2.0 ** int(rng.integers(-40, 10)))) for _ in range(60)]
    for w in ws:
        top = 1 << 19
        ks = np.concatenate([np.arange(0, 300), rng.integers(0, top, size=400),
                             [(1 << p) + d for p in range(1, 19) for d in (-2, -1, 0, 1, 2)], [top]])
        assert _check(lib, w, ks) == 0, w


def test_long_walks_and_stagnation(lib):
    # 2^26 steps: the sums of 0.1 stagnate at 2^21 = 2097152 (0.1 is below half an ulp there) -- the jump must say so too
    for w in (0.1, 1.3):
        ks = [1 << 20, (1 << 24) - 1, 1 << 24, (1 << 24) + 1, 1 << 25, 1 << 26]
        assert _check(lib, w, ks) == 0, w
    assert lib.seqsum_jump(ctypes.c_float(0.1), 1 << 40) == lib.seqsum_jump(ctypes.c_float(0.1), 1 << 27)
    # a huge index costs no more than a small one
    assert np.isfinite(lib.seqsum_jump(ctypes.c_float(1.3), (1 << 62)))


def test_degenerate_voxel_sizes_need_no_long_walk(lib):
    """NaN, infinities, zero, negative and subnormal voxel sizes: the jump conditions cannot serve them; every thread of
    the table kernel would otherwise walk its whole prefix one addition at a time.  Values: the plain loop's."""
    import time
    for w in (-0.1, -1.3, -6.0, 0.0, 2.0 ** -149, 2.0 ** -149 * 5, 2.0 ** -140, 1.0e-39, -1.0e-39, 2.0 ** -127):
        ks = np.concatenate([np.arange(0, 200), [1 << 12, (1 << 15) + 3, 1 << 18]])
        assert _check(lib, w, ks) == 0, w
    t0 = time.perf_counter()
    for w in (float("nan"), float("inf"), float("-inf"), 0.0, -0.1, 2.0 ** -149, 1.0e-39, -2.0 ** -140, 3.0e38):
        v = lib.seqsum_jump(ctypes.c_float(w), 1 << 40)
        if w != w:
            assert v != v
        elif w == 0.0:
            assert v == 0.0
        elif abs(w) == float("inf") or abs(w) > 1e38:
            assert v == (float("inf") if w > 0 else float("-inf"))
    assert time.perf_counter() - t0 < 2.0, "a degenerate voxel size walked a 2^40-step line"
    # the subnormal sums: exact multiples until they reach the normal range
    assert lib.seqsum_jump(ctypes.c_float(2.0 ** -149), 1000) == np.float32(1000 * 2.0 ** -149)
