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
