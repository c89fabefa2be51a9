"""CPU tier: the 16-bit integer column pass (csrc/edt_colq16_lane.h), emulated lane by lane.

tests/q16_emul.cpp compiles the SAME per-lane header the HIP kernel is built from with g++ and plays the fill, the scans,
the break bits and every block of every tile.  Checked bit for bit against the oracle on 2-D images: pass 1 comes from the
oracle's 1-D transform (as fp32 values, or as the 16-bit distance indices of the index form), the emulation supplies
pass 2, the oracle's 2-D transform is the expected result.  Tiles that do not qualify (values beyond 16 bits, values off
the quantum grid) must be refused, never written.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from synth import blocky_labels

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "euclidean-distance-transform-3d_amd", "csrc")
BUILD = os.path.join(ROOT, "tests", "_build")
FLT_MAX = np.float32(3.402823466e+38)


@pytest.fixture(scope="module")
def q16():
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "libq16_emul.so")
    src = os.path.join(ROOT, "tests", "q16_emul.cpp")
    hdr = os.path.join(CSRC, "edt_colq16_lane.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        tmp = f"{so}.{os.getpid()}.tmp"
        subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC",
                        f"-I{CSRC}", src, "-o", tmp], check=True)
        os.replace(tmp, so)
    lib = ctypes.CDLL(so)
    lib.q16_emul_column_pass.restype = ctypes.c_int
    lib.q16_emul_quantum.restype = ctypes.c_int
    lib.q16_emul_set_no_wide(0)
    lib.q16_emul_set_full_wide(0)
    return lib


def wide_limit(a, q, n=None, bb=True, with_inf=False):
    """largest N of the wide form (edt_colq16_lane.h: q16_dmax_wide, q16_wide_range): N * odd(q) < 2^24, N <= a * d^2 for a
    d <= 2047; without a black border (n given) lowered so that N + a * n^2 stays exact, where columns of n rows may carry +inf
    at all (with_inf: also return whether they may)"""
    m, _ = np.frexp(np.float64(q))
    m = int(m * (1 << 24))
    while m % 2 == 0:
        m //= 2
    cap = ((1 << 24) - 1) // m
    d = 1
    while d < 2047 and a * (d + 1) * (d + 1) <= cap:
        d += 1
    lim = a * d * d if a * d * d <= cap else 0
    d16 = int(np.floor(np.sqrt(65534 / a)))
    inf_ok = False
    if not bb and n is not None and lim > a * d16 * d16 and n <= d and a * n * n < cap and cap - a * n * n > a * d16 * d16:
        lim = min(lim, cap - a * n * n)
        inf_ok = True
    return (lim, inf_ok) if with_inf else lim


def quantum(lib, w):
    wa = (ctypes.c_float * 3)(*[float(v) for v in w], *([1.0] * (3 - len(w))))
    q = ctypes.c_float(0)
    a = (ctypes.c_uint32 * 3)()
    ok = lib.q16_emul_quantum(wa, ctypes.c_int(len(w)), ctypes.byref(q), a)
    return bool(ok), q.value, [int(v) for v in a]


def column_pass(lib, labels_yx, f_yx, codes_yx, q, a, ain, bb, epi):
    n, sx = labels_yx.shape
    lab = np.ascontiguousarray(labels_yx, dtype=np.uint32)
    out = np.full((n, sx), -1.0, dtype=np.float32)
    ok = np.zeros((sx + 31) // 32, dtype=np.uint8)
    f = None if f_yx is None else np.ascontiguousarray(f_yx, dtype=np.float32)
    c = None if codes_yx is None else np.ascontiguousarray(codes_yx, dtype=np.uint16)
    lib.q16_emul_column_pass(lab.ctypes.data_as(ctypes.c_void_p),
                             f.ctypes.data_as(ctypes.c_void_p) if f is not None else None,
                             c.ctypes.data_as(ctypes.c_void_p) if c is not None else None,
                             out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(sx), ctypes.c_int64(n),
                             ctypes.c_float(q), ctypes.c_uint32(a), ctypes.c_uint32(ain), ctypes.c_int(int(bb)),
                             ctypes.c_int(epi), ok.ctypes.data_as(ctypes.c_void_p))
    # per x-tile: 0 = handed to the fp32 kernel, 1 = 16-bit form, 2 = wide form (two passes of 16 columns, 32-bit lanes),
    # 3 = 16-bit form + ONE wide pass over the (at most 16) columns that hold values beyond 16 bits
    return out, ok


def x_pass(oracle, labels_yx, wx, bb):
    rows = [oracle.raw1d(np.ascontiguousarray(r, dtype=np.uint32), 2, r.size, wx, bb) for r in labels_yx]
    f = np.stack(rows).astype(np.float32)
    codes = np.where(np.isinf(f), 0xFFFF, np.rint(np.sqrt(f.astype(np.float64)) / wx)).astype(np.uint16)
    if not bb:
        f[np.isinf(f)] = FLT_MAX  # tofinite (src/edt.hpp:39-45)
    return f, codes


def make_labels(n, sx, kind, rng):
    if kind == "ones":
        return np.ones((n, sx), dtype=np.uint32)
    if kind == "blocky":
        return blocky_labels((n, sx), nlabels=4, zero_frac=0.15, block=int(rng.integers(3, 40)), rng=rng).astype(np.uint32)
    if kind == "noise":
        return rng.integers(0, 3, size=(n, sx)).astype(np.uint32)
    if kind == "cells":
        return blocky_labels((n, sx), nlabels=200, zero_frac=0.0, block=int(rng.integers(20, 90)), rng=rng).astype(np.uint32)
    lab = blocky_labels((n, sx), nlabels=2, zero_frac=0.0, block=int(rng.integers(20, 200)), rng=rng).astype(np.uint32)
    lab[rng.random((n, sx)) < 0.01] = 0
    return lab


CASES = []
for n, sx in ((1024, 32), (1000, 36), (900, 64), (513, 8), (512, 64), (512, 96), (500, 36), (300, 40), (257, 40), (256, 32),
              (130, 96), (128, 8), (100, 44), (64, 32), (33, 64), (32, 4), (17, 12), (8, 8), (1, 8)):
    for kind in ("ones", "blocky", "noise", "membrane", "cells"):
        CASES.append((n, sx, kind))


@pytest.mark.parametrize("n,sx,kind", CASES)
def test_q16_column_pass_matches_oracle(q16, oracle_port, n, sx, kind):
    rng = np.random.default_rng(n * 1000 + sx)
    lab = make_labels(n, sx, kind, rng)
    for (wx, wy) in ((1.0, 1.0), (6.0, 30.0), (30.0, 6.0), (0.5, 1.0), (4.0, 40.0)):
        ok, q, a = quantum(q16, (wx, wy))
        assert ok, (wx, wy)
        assert np.float32(q) * a[0] == np.float32(wx) ** 2 and np.float32(q) * a[1] == np.float32(wy) ** 2
        for bb in (True, False):
            f1, codes = x_pass(oracle_port, lab, wx, bb)
            want = oracle_port.raw2d(lab, 2, sx, n, (wx, wy), bb).reshape(n, sx)
            for form in ("f32", "codes"):
                # (epi bit 0: the pass is the transform's last one and black_border is off -- rows without any boundary, +inf in
                # the integer kernel's wide form, leave as +INF: toinfinite, src/edt.hpp:47-53)
                inf = 0 if bb else 1
                got, tiles = column_pass(q16, lab, f1 if form == "f32" else None, codes if form == "codes" else None,
                                         q, a[1], a[0], bb, inf)
                got_s, tiles_s = column_pass(q16, lab, f1 if form == "f32" else None, codes if form == "codes" else None,
                                             q, a[1], a[0], bb, 2 | inf)
                assert np.array_equal(tiles, tiles_s)
                for i, t_ok in enumerate(tiles):
                    sl = slice(32 * i, min(sx, 32 * i + 32))
                    if t_ok:
                        assert np.array_equal(got[:, sl], want[:, sl]), (n, sx, kind, wx, wy, bb, form, i)
                        assert np.array_equal(got_s[:, sl], np.sqrt(want[:, sl])), (n, sx, kind, wx, wy, bb, form, i, "sqrt")
                    else:
                        assert (got[:, sl] == -1.0).all(), "a refused tile was written"
                        # a refusal has a reason: a value beyond the tile limit -- of the wide form, where there is one -- (rows
                        # without a boundary included)
                        dmax = int(np.floor(np.sqrt(65534 / a[1])))
                        wl, inf_ok = wide_limit(a[1], q, n, bb, with_inf=True)
                        lim = max(a[1] * dmax * dmax, wl)
                        fin = f1[:, sl][f1[:, sl] < np.float32(3e38)].astype(np.float64)
                        kfin = codes[:, sl][codes[:, sl] != 0xFFFF].astype(np.int64)
                        assert (fin.size and (fin / q).max() > lim) or (kfin.size and (kfin ** 2 * a[0]).max() > lim) or \
                            (not inf_ok and (codes[:, sl] == 0xFFFF).any())
                # (without a black border a row inside one label has no boundary at all: FLT_MAX, the tile is refused)
                if bb and (kind in ("blocky", "noise", "cells") or (kind == "membrane" and n <= 512)) and wx <= 6.0:
                    assert tiles.any(), (n, sx, kind, wx, wy, bb, form)


def test_q16_refuses_values_off_the_quantum_grid(q16, oracle_port):
    rng = np.random.default_rng(5)
    lab = make_labels(128, 64, "cells", rng)
    f1, _ = x_pass(oracle_port, lab, 1.0, True)
    f1[40, 5] = np.float32(2.5)           # not a multiple of q = 1
    f1[90, 40] = np.float32(70000.0)      # beyond 16 bits: the wide form
    f1[91, 41] = np.float32(70000.5)      # ... and off the grid
    _, tiles = column_pass(q16, lab, f1, None, 1.0, 1, 1, True, 0)
    assert list(tiles) == [0, 0]
    f1[40, 5] = np.float32(2.0)
    f1[91, 41] = np.float32(2047 ** 2 + 1)    # beyond the wide form (a = 1: N <= 2047^2; never more than 2^24 - 1)
    _, tiles = column_pass(q16, lab, f1, None, 1.0, 1, 1, True, 0)
    assert list(tiles) == [1, 0]
    f1[91, 41] = np.float32(2047 ** 2)
    _, tiles = column_pass(q16, lab, f1, None, 1.0, 1, 1, True, 0)
    assert list(tiles) == [1, 3]              # (one column beyond 16 bits: that column alone gets a wide pass)
    q16.q16_emul_set_full_wide(1)
    _, tiles = column_pass(q16, lab, f1, None, 1.0, 1, 1, True, 0)
    q16.q16_emul_set_full_wide(0)
    assert list(tiles) == [1, 2]
    q16.q16_emul_set_no_wide(1)           # (the 16-bit form alone, as in round 4)
    _, tiles = column_pass(q16, lab, f1, None, 1.0, 1, 1, True, 0)
    q16.q16_emul_set_no_wide(0)
    assert list(tiles) == [1, 0]


WIDE_CASES = [(64, 640, "ones"), (300, 1100, "ones"), (700, 600, "ones"), (520, 560, "bigcells"), (1024, 1200, "bigcells"),
              (200, 2048, "ones"), (96, 900, "sparse")]


@pytest.mark.parametrize("n,sx,kind", WIDE_CASES)
def test_q16_wide_form_matches_oracle(q16, oracle_port, n, sx, kind):
    """Tiles that hold values beyond 16 bits (the middle of rows of more than 510 voxels, objects deeper than ~255 voxels) are
    worked on as two half-tiles with 32-bit lanes -- the same lane code (V<true>) -- instead of being handed to the fp32
    kernel: bit-identical to the oracle, both input forms, both borders, sqrt; and they do exist in these cases."""
    rng = np.random.default_rng(n + 7 * sx)
    if kind == "ones":
        lab = np.ones((n, sx), dtype=np.uint32)
    elif kind == "bigcells":
        lab = blocky_labels((n, sx), nlabels=3, zero_frac=0.0, block=int(rng.integers(280, 420)), rng=rng).astype(np.uint32)
        lab[rng.random((n, sx)) < 0.0002] = 0
    else:
        lab = np.ones((n, sx), dtype=np.uint32)
        lab[rng.random((n, sx)) < 0.0005] = 0
    seen_wide = seen_subset = False
    for (wx, wy) in ((1.0, 1.0), (6.0, 30.0), (30.0, 6.0), (0.5, 1.0)):
        ok, q, a = quantum(q16, (wx, wy))
        assert ok
        for bb in (True, False):
            f1, codes = x_pass(oracle_port, lab, wx, bb)
            want = oracle_port.raw2d(lab, 2, sx, n, (wx, wy), bb).reshape(n, sx)
            for form in ("f32", "codes"):
                for epi, full in ((0, 0), (2, 0), (0, 1)):   # (full: no column subsets -- two wide passes over every such tile)
                    q16.q16_emul_set_full_wide(full)
                    got, tiles = column_pass(q16, lab, f1 if form == "f32" else None, codes if form == "codes" else None,
                                             q, a[1], a[0], bb, epi | (0 if bb else 1))
                    q16.q16_emul_set_full_wide(0)
                    assert not full or not (tiles == 3).any()
                    exp = np.sqrt(want) if epi else want
                    for i, t_ok in enumerate(tiles):
                        sl = slice(32 * i, min(sx, 32 * i + 32))
                        if t_ok:
                            assert np.array_equal(got[:, sl], exp[:, sl]), (n, sx, kind, wx, wy, bb, form, epi, i, int(t_ok))
                        else:
                            assert (got[:, sl] == -1.0).all()
                    seen_wide |= bool((tiles == 2).any())
                    seen_subset |= bool((tiles == 3).any())
                    if bb and kind == "ones" and int(codes.max()) ** 2 * a[0] <= wide_limit(a[1], q):
                        assert tiles.all(), "a single label inside a black border leaves nothing to the fp32 kernel"
    assert seen_wide and (seen_subset or kind == "bigcells")


@pytest.mark.parametrize("n,sx", [(1024, 64), (600, 96), (300, 40), (200, 64), (97, 32)])
def test_q16_rows_without_boundary(q16, oracle_port, n, sx):
    """No black border, rows of one label from edge to edge: +inf after pass X.  Along the column such a row finds a border far
    away (a * d^2 for any d up to the column's length: no clamp of the distance may apply) or a finite site of a neighbouring
    row (a sum N[j] + a * d^2 beyond every value the tile held): carried by the wide form where the column is short enough
    for both to stay exact (q16_wide_range), refused otherwise -- and never wrong."""
    rng = np.random.default_rng(n + sx)
    lab = np.ones((n, sx), dtype=np.uint32)
    lab[:2] = 2                                     # a border along the column, far from most rows
    lab[n // 3, 3 * sx // 4:] = 0                   # one row with a boundary: finite sites for the rows around it
    lab[n - 5:, :sx // 2] = 3
    lab[rng.integers(0, n, 3), rng.integers(0, sx, 3)] = 0
    carried = refused = False
    for (wx, wy) in ((1.0, 1.0), (6.0, 30.0), (30.0, 6.0), (0.5, 1.0), (2.0, 40.0), (40.0, 2.0)):
        ok, q, a = quantum(q16, (wx, wy))
        assert ok
        f1, codes = x_pass(oracle_port, lab, wx, False)
        assert (codes == 0xFFFF).any()
        want = oracle_port.raw2d(lab, 2, sx, n, (wx, wy), False).reshape(n, sx)
        wl, inf_ok = wide_limit(a[1], q, n, False, with_inf=True)
        for form in ("f32", "codes"):
            for epi, full in ((1, 0), (3, 0), (1, 1), (0, 0)):
                q16.q16_emul_set_full_wide(full)
                got, tiles = column_pass(q16, lab, f1 if form == "f32" else None, codes if form == "codes" else None,
                                         q, a[1], a[0], False, epi)
                q16.q16_emul_set_full_wide(0)
                exp = np.sqrt(want) if epi & 2 else want
                if not epi & 1:
                    exp = np.where(np.isinf(exp), np.finfo(np.float32).max, exp)  # (between the passes: FLT_MAX)
                for i, t_ok in enumerate(tiles):
                    sl = slice(32 * i, min(sx, 32 * i + 32))
                    if t_ok:
                        assert np.array_equal(got[:, sl], exp[:, sl]), (n, sx, wx, wy, form, epi, full, i, int(t_ok))
                        assert inf_ok or not (codes[:, sl] == 0xFFFF).any()
                        carried |= bool((codes[:, sl] == 0xFFFF).any())
                    else:
                        assert (got[:, sl] == -1.0).all()
                        refused = True
    assert carried and (refused or n < 280)


@pytest.mark.parametrize("n,sx,kind", [(512, 64, "one_row"), (700, 32, "one_voxel"), (512, 96, "stretches"), (1000, 64, "stretches"),
                                       (333, 64, "top_and_bottom"), (512, 32, "two_rows"), (200, 96, "stretches"), (97, 32, "one_voxel")])
def test_q16_windows_over_rows_without_boundary(q16, oracle_port, n, sx, kind):
    """Round 6 (the window of a +inf row): without a black border a column may hold ONE finite row, or finite
    stretches between long stretches of rows that saw no boundary along x.  The wide form starts such a window where the
    stretch of +inf rows ends (Block::dskip) and ends it where the column's finite rows end (Block::dend, finite_extent)
    instead of walking hundreds of steps over +inf -- the results must stay the oracle's for every structure: one finite row,
    one finite voxel, finite rows at the column's ends (no break there), alternating stretches, every voxel-size pair."""
    rng = np.random.default_rng(7 * n + sx)
    lab = np.ones((n, sx), dtype=np.uint32)   # (rows, x): the column pass runs along axis 0
    if kind == "one_row":
        lab[n // 2, sx // 3] = 0
    elif kind == "one_voxel":
        lab[n // 2, sx // 3] = 0
        lab[: n // 5] = 2                       # and a label border along the column, far from most rows
    elif kind == "two_rows":
        lab[n // 6, 5] = 0
        lab[n - 1 - n // 7, sx - 3] = 0
    elif kind == "top_and_bottom":
        lab[0, sx // 2] = 0                     # finite rows that touch the column's ends: no break marks their outer side
        lab[n - 1, 3] = 0
        lab[n // 2, sx - 1] = 5
    else:                                       # alternating stretches of rows with and without a boundary
        r = 0
        while r < n:
            ln = int(rng.integers(3, max(4, n // 4)))
            if rng.random() < 0.5:
                lab[r:r + ln, int(rng.integers(0, sx))] = int(rng.integers(2, 6))   # a second label somewhere in these rows
            r += ln + int(rng.integers(0, max(2, n // 3)))
        lab[int(rng.integers(0, n)), :] = 7     # a whole row of another label: +inf along x, a border along the column
    seen_inf = False
    for (wx, wy) in ((1.0, 1.0), (6.0, 30.0), (30.0, 6.0), (0.5, 1.0), (2.0, 1.0)):
        ok, q, a = quantum(q16, (wx, wy))
        assert ok
        f1, codes = x_pass(oracle_port, lab, wx, False)
        assert (codes == 0xFFFF).any()
        want = oracle_port.raw2d(lab, 2, sx, n, (wx, wy), False).reshape(n, sx)
        for form in ("f32", "codes"):
            for epi, full in ((1, 0), (3, 1), (0, 0)):
                q16.q16_emul_set_full_wide(full)
                got, tiles = column_pass(q16, lab, f1 if form == "f32" else None, codes if form == "codes" else None,
                                         q, a[1], a[0], False, epi)
                q16.q16_emul_set_full_wide(0)
                exp = np.sqrt(want) if epi & 2 else want
                if not epi & 1:
                    exp = np.where(np.isinf(exp), np.finfo(np.float32).max, exp)
                for i, t_ok in enumerate(tiles):
                    sl = slice(32 * i, min(sx, 32 * i + 32))
                    if t_ok:
                        assert np.array_equal(got[:, sl], exp[:, sl]), (n, sx, kind, wx, wy, form, epi, full, i, int(t_ok))
                        seen_inf |= bool((codes[:, sl] == 0xFFFF).any())
                    else:
                        assert (got[:, sl] == -1.0).all()
    assert seen_inf or n > 280


@pytest.mark.parametrize("n,sx,kind", [(1024, 32, "cells"), (512, 64, "blocky"), (500, 36, "membrane"), (300, 40, "cells"),
                                       (130, 96, "noise"), (257, 40, "blocky"), (64, 32, "cells"), (33, 8, "ones"), (16, 8, "blocky")])
def test_q16_output_stride_two(q16, oracle_port, n, sx, kind):
    """blocks of 16 rows whose even rows are evaluated and written (every row a candidate): the doubled grids of the
    voxel-graph transform.  The even rows must be the oracle's, the odd ones untouched."""
    rng = np.random.default_rng(n + sx)
    lab = make_labels(n, sx, kind, rng)
    for (wx, wy) in ((1.0, 1.0), (3.0, 15.0), (0.5, 0.5)):
        ok, q, a = quantum(q16, (wx, wy))
        assert ok
        for bb in (True, False):
            f1, _ = x_pass(oracle_port, lab, wx, bb)
            want = oracle_port.raw2d(lab, 2, sx, n, (wx, wy), bb).reshape(n, sx)
            for epi in (0, 2):
                out = np.full((n, sx), -1.0, dtype=np.float32)
                tiles = np.zeros((sx + 31) // 32, dtype=np.uint8)
                labc, fc = np.ascontiguousarray(lab, dtype=np.uint32), np.ascontiguousarray(f1)
                q16.q16_emul_column_pass_even(labc.ctypes.data_as(ctypes.c_void_p), fc.ctypes.data_as(ctypes.c_void_p),
                                              out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(sx), ctypes.c_int64(n),
                                              ctypes.c_float(q), ctypes.c_uint32(a[1]), ctypes.c_int(int(bb)), ctypes.c_int(epi),
                                              tiles.ctypes.data_as(ctypes.c_void_p))
                exp = np.sqrt(want) if epi else want
                for i, t_ok in enumerate(tiles):
                    sl = slice(32 * i, min(sx, 32 * i + 32))
                    if t_ok:
                        assert np.array_equal(out[0::2, sl], exp[0::2, sl]), (n, sx, kind, wx, wy, bb, epi, i)
                    assert (out[1::2, sl] == -1.0).all()
                if bb:
                    assert tiles.any()


def test_q16_plane_between_passes(q16, oracle_port):
    """the results of a pass stay 16-bit (plane out), the next pass takes every row from the plane or from fp32 values
    (mixed input); a refused tile of the mixed form leaves its plane rows as fp32 values for the fp32 kernel"""
    rng = np.random.default_rng(11)
    for (n, sx, kind) in ((512, 64, "cells"), (300, 40, "blocky"), (130, 96, "membrane")):
        lab = make_labels(n, sx, kind, rng)
        for (wx, wy) in ((1.0, 1.0), (6.0, 30.0)):
            ok, q, a = quantum(q16, (wx, wy))
            f1, codes = x_pass(oracle_port, lab, wx, True)
            want = oracle_port.raw2d(lab, 2, sx, n, (wx, wy), True).reshape(n, sx)
            labc = np.ascontiguousarray(lab, dtype=np.uint32)
            tiles = np.zeros((sx + 31) // 32, dtype=np.uint8)
            # plane out: N = result / q as 16-bit integers
            plane = np.full((n, sx), 0xABCD, dtype=np.uint16)
            cc = np.ascontiguousarray(codes)
            q16.q16_emul_column_pass_plane(labc.ctypes.data_as(ctypes.c_void_p), None, cc.ctypes.data_as(ctypes.c_void_p), None,
                                           ctypes.c_int64(sx), ctypes.c_int64(n), ctypes.c_float(q), ctypes.c_uint32(a[1]),
                                           ctypes.c_uint32(a[0]), 1, 0, tiles.ctypes.data_as(ctypes.c_void_p), None, None,
                                           plane.ctypes.data_as(ctypes.c_void_p))
            assert tiles.all()
            assert np.array_equal(plane.astype(np.float32) * np.float32(q), want)
            # mixed input: a random half of the rows from a plane holding f1 / q, the others as fp32 values
            rows = (rng.random(n) < 0.5).astype(np.uint8)
            pin = np.where(rows[:, None] == 1, np.rint(f1 / np.float32(q)), 0xEEEE).astype(np.uint16)
            fin = np.where(rows[:, None] == 1, np.float32(-7.0), f1).astype(np.float32)
            out = np.full((n, sx), -1.0, dtype=np.float32)
            q16.q16_emul_column_pass_plane(labc.ctypes.data_as(ctypes.c_void_p), fin.ctypes.data_as(ctypes.c_void_p), None,
                                           out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(sx), ctypes.c_int64(n),
                                           ctypes.c_float(q), ctypes.c_uint32(a[1]), ctypes.c_uint32(a[0]), 1, 0,
                                           tiles.ctypes.data_as(ctypes.c_void_p), pin.ctypes.data_as(ctypes.c_void_p),
                                           rows.ctypes.data_as(ctypes.c_void_p), None)
            assert tiles.all() and np.array_equal(out, want), (n, sx, kind, wx, wy)
            # a value off the grid in an fp32 row: the tile is refused, its plane rows arrive as fp32 values
            r0 = int(np.flatnonzero(rows == 0)[0])
            fin[r0, 3] = np.float32(2.5) * np.float32(q)
            out[:] = -1.0
            q16.q16_emul_column_pass_plane(labc.ctypes.data_as(ctypes.c_void_p), fin.ctypes.data_as(ctypes.c_void_p), None,
                                           out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(sx), ctypes.c_int64(n),
                                           ctypes.c_float(q), ctypes.c_uint32(a[1]), ctypes.c_uint32(a[0]), 1, 0,
                                           tiles.ctypes.data_as(ctypes.c_void_p), pin.ctypes.data_as(ctypes.c_void_p),
                                           rows.ctypes.data_as(ctypes.c_void_p), None)
            assert not tiles[0]
            assert np.array_equal(out[rows == 1, :32], f1[rows == 1, :32]) and (out[rows == 0, :32] == -1.0).all()


def test_q16_refused_tile_turns_inf_rows_of_the_plane_into_flt_max(q16, oracle_port):
    """Round 6 (found by the closing fuzz on the GPU): a row of the 16-bit plane may be +inf (0xFFFF: a tile of nothing but +inf
    that the pass before left there); a tile that is REFUSED hands its plane rows to the fp32 kernel as fp32 values -- +inf as
    FLT_MAX (tofinite, src/edt.hpp:39-45), not as 65535 quanta.  And a tile that is not refused carries those rows as +inf."""
    n, sx = 200, 32
    lab = np.ones((n, sx), dtype=np.uint32)
    labc = np.ascontiguousarray(lab)
    ok, q, a = quantum(q16, (1.0, 1.0))
    rows = np.ones(n, dtype=np.uint8)
    rows[50] = 0                                           # one row arrives as fp32 values ...
    pin = np.full((n, sx), 0xFFFF, dtype=np.uint16)        # ... every other row is +inf in the plane
    fin = np.full((n, sx), -7.0, dtype=np.float32)
    fin[50] = np.float32(9.0)                              # a finite row: 9 quanta
    tiles = np.zeros(1, dtype=np.uint8)
    out = np.full((n, sx), -1.0, dtype=np.float32)
    args = lambda: (labc.ctypes.data_as(ctypes.c_void_p), fin.ctypes.data_as(ctypes.c_void_p), None, out.ctypes.data_as(ctypes.c_void_p),
                    ctypes.c_int64(sx), ctypes.c_int64(n), ctypes.c_float(q), ctypes.c_uint32(a[1]), ctypes.c_uint32(a[0]), 0, 1,
                    tiles.ctypes.data_as(ctypes.c_void_p), pin.ctypes.data_as(ctypes.c_void_p), rows.ctypes.data_as(ctypes.c_void_p), None)
    q16.q16_emul_column_pass_plane(*args())
    assert tiles[0]                                        # carried: every row finds the finite one, (p - 50)^2 + 9
    p = np.arange(n, dtype=np.float32)[:, None]
    assert np.array_equal(out, np.broadcast_to((p - 50) ** 2 + 9, (n, sx)))
    fin[50, 3] = np.float32(2.5)                           # off the quantum grid: the tile is refused
    out[:] = -1.0
    q16.q16_emul_column_pass_plane(*args())
    assert not tiles[0]
    assert (out[rows == 1] == FLT_MAX).all() and (out[50] == -1.0).all()


def test_quantum_of_voxel_sizes(q16):
    assert quantum(q16, (1.0, 1.0, 1.0)) == (True, 1.0, [1, 1, 1])
    assert quantum(q16, (6.0, 6.0, 30.0)) == (True, 36.0, [1, 1, 25])
    assert quantum(q16, (4.0, 4.0, 40.0)) == (True, 16.0, [1, 1, 100])
    assert quantum(q16, (0.5, 0.5, 1.0)) == (True, 0.25, [1, 1, 4])
    assert quantum(q16, (3.0, 5.0)) == (True, 1.0, [9, 25, 1])
    ok, q, a = quantum(q16, (2.0 ** -30, 2.0 ** -30, 2.0 ** -29))
    assert ok and a == [1, 1, 4] and q == 2.0 ** -60
    for w in ((3.58, 3.58, 40.0), (1.1, 1.1, 1.1), (0.7, 1.3), (1.0, 1.0, 1000.0), (1.0, 1e-3), (16381.0, 1.0), (1.0, 0.0),
              (float("nan"), 1.0), (1.0, float("inf"))):
        assert not quantum(q16, w)[0], w
    # every quantum reproduces the squares exactly, and N * q is exact for 16-bit N
    rng = np.random.default_rng(1)
    for _ in range(300):
        w = [float(np.float32(rng.integers(1, 64) * 2.0 ** int(rng.integers(-6, 6)))) for _ in range(3)]
        ok, q, a = quantum(q16, w)
        if not ok:
            continue
        for wi, ai in zip(w, a):
            assert np.float32(q) * np.float32(ai) == np.float32(wi) * np.float32(wi)
        for N in (1, 3, 65533, 65534):
            assert float(np.float32(N) * np.float32(q)) == N * float(q)


def host_cannot_refuse(a, q, shape_xyz, bb):
    """csrc/edt_api.hip: q16_cannot_refuse, restated -- (pass Y, pass Z): the host's proof that the integer kernel refuses no
    tile of the pass, on which it skips the fp32 launch over the hand-over list"""
    sx, sy, sz = shape_xyz
    kmax = (sx + 1) // 2 if bb else sx
    vmax_x = kmax * kmax * a[0]

    def limit(ai, n):
        d16 = int(np.floor(np.sqrt(65534 / ai)))
        wl, inf_ok = wide_limit(ai, q, n, bb, with_inf=True)
        if not bb:
            return wl if inf_ok else 0
        return max(ai * d16 * d16, wl)
    y = vmax_x <= limit(a[1], sy)
    z = y and vmax_x + (0 if bb else sy * sy * a[1]) <= limit(a[2], sz)
    return y, z


@pytest.mark.parametrize("shape,kind", [((64, 120, 97), "ones"), ((64, 120, 97), "slabs"), ((96, 200, 130), "slabs"),
                                        ((32, 413, 216), "blocky"), ((64, 300, 100), "bigblocks"), ((128, 97, 140), "ones"),
                                        ((64, 140, 260), "slabs"), ((32, 300, 300), "slabs")])
def test_q16_three_passes_and_the_hosts_proof(q16, oracle_port, shape, kind):
    """Passes Y and Z of a 3-D volume through the lane logic (pass X from the oracle), both borders: every tile the kernel
    accepts is bit-identical to the oracle -- rows without any boundary (+inf after X, FLT_MAX between the passes, +INF at
    the end) included -- and wherever the host's proof says that no tile can be refused, none is."""
    sx, sy, sz = shape
    rng = np.random.default_rng(sx + sy + sz)
    if kind == "ones":
        lab = np.ones((sz, sy, sx), dtype=np.uint32)
    elif kind == "slabs":
        lab = np.ones((sz, sy, sx), dtype=np.uint32)
        lab[:, :3, :] = 2            # borders along Y far from most rows
        lab[:2, :, : sx // 2] = 3    # and along Z
        lab[sz // 2, sy // 2, sx // 3] = 0
    else:
        lab = blocky_labels((sz, sy, sx), nlabels=4, zero_frac=0.05, block=30 if kind == "blocky" else 90, rng=rng).astype(np.uint32)
    proved = unproved = 0
    # ((1, 10, 10) on 300 x 300 columns: pass Y carries +inf and turns it into values that pass Z's range does not hold)
    sizes = ((1.0, 1.0, 1.0), (6.0, 6.0, 30.0), (0.5, 40.0, 2.0), (6.0, 40.0, 3.0), (30.0, 6.0, 2.0), (1.0, 10.0, 10.0))
    if shape == (32, 300, 300):
        sizes = ((1.0, 1.0, 1.0), (1.0, 10.0, 10.0))
    elif shape == (32, 413, 216):   # (the two mismatches of the GPU fuzz on the first version of "+inf in the wide form")
        sizes = ((1.0, 1.0, 1.0), (0.5, 40.0, 2.0), (6.0, 40.0, 3.0))
    for w in sizes:
        ok, q, a = quantum(q16, w)
        assert ok
        for bb in (True, False):
            sure_y, sure_z = host_cannot_refuse(a, q, shape, bb)
            assert library_proof(shape, w, bb) == (1, sure_y, sure_z)   # (what run_device acts on)
            want = oracle_port.raw3d(np.ascontiguousarray(lab).reshape(-1), 2, sx, sy, sz, w, bb).reshape(sz, sy, sx)
            after_y = np.empty((sz, sy, sx), dtype=np.float32)
            for z in range(sz):
                _, codes = x_pass(oracle_port, lab[z], w[0], bb)
                w2 = oracle_port.raw2d(np.ascontiguousarray(lab[z]).reshape(-1), 2, sx, sy, (w[0], w[1]), bb).reshape(sy, sx)
                w2 = np.where(np.isinf(w2), np.float32(FLT_MAX), w2).astype(np.float32)
                got, tiles = column_pass(q16, lab[z], None, codes, q, a[1], a[0], bb, 0)
                assert not sure_y or tiles.all(), (shape, kind, w, bb, z, "the host's proof for pass Y")
                for i, t_ok in enumerate(tiles):
                    sl = slice(32 * i, min(sx, 32 * i + 32))
                    if t_ok:
                        assert np.array_equal(got[:, sl], w2[:, sl]), (shape, kind, w, bb, z, i, int(t_ok))
                after_y[z] = w2
            for y in range(sy):
                got, tiles = column_pass(q16, lab[:, y, :], after_y[:, y, :], None, q, a[2], a[0], bb, 0 if bb else 1)
                assert not sure_z or tiles.all(), (shape, kind, w, bb, y, "the host's proof for pass Z")
                for i, t_ok in enumerate(tiles):
                    sl = slice(32 * i, min(sx, 32 * i + 32))
                    if t_ok:
                        assert np.array_equal(got[:, sl], want[:, y, sl]), (shape, kind, w, bb, y, i, int(t_ok))
            proved += int(sure_z)
            unproved += int(not sure_z)
    assert proved and unproved


def library_proof(shape_xyz, w, bb):
    """the product library's own answer (include/edt_hip.h: edt_hip_q16_no_refusals -- host arithmetic, no device)"""
    from edt import _lib
    lib = _lib.load()
    y, z = ctypes.c_int(-1), ctypes.c_int(-1)
    ndim = len(shape_xyz)
    sx, sy = shape_xyz[0], shape_xyz[1]
    sz = shape_xyz[2] if ndim == 3 else 1
    ws = [float(v) for v in w] + [1.0] * (3 - len(w))
    rc = lib.edt_hip_q16_no_refusals(sx, sy, sz, ws[0], ws[1], ws[2], ndim, int(bb), ctypes.addressof(y), ctypes.addressof(z))
    return rc, bool(y.value), bool(z.value)


def test_the_librarys_proof_is_the_one_the_lane_logic_was_held_against(q16):
    """host_cannot_refuse above -- the restatement test_q16_three_passes_and_the_hosts_proof checks against what the tiles of
    the lane logic do -- and the library's own decision (csrc/edt_colq16.hip: q16_no_refusals, what run_device skips the fp32
    launch on) agree on random extents, voxel sizes and borders, and on the cases of that test."""
    rng = np.random.default_rng(77)
    cases = [((sx, sy, sz), w, bb) for (sx, sy, sz) in ((64, 120, 97), (32, 413, 216), (32, 300, 300), (64, 140, 260), (512, 512, 512),
                                                        (1024, 1024, 1024), (72, 518, 352))
             for w in ((1.0, 1.0, 1.0), (6.0, 6.0, 30.0), (0.5, 40.0, 2.0), (6.0, 40.0, 3.0), (1.0, 10.0, 10.0), (4.0, 4.0, 40.0))
             for bb in (True, False)]
    for _ in range(3000):
        shape = (int(rng.integers(1, 4097)), int(rng.integers(97, 1025)), int(rng.integers(97, 1025)))
        w = tuple(float(v) for v in rng.choice([1, 2, 6, 30, 0.5, 4, 40, 3, 10, 0.25, 7.25, 1.3], size=3))
        cases.append((shape, w, bool(rng.integers(0, 2))))
    seen = set()
    for shape, w, bb in cases:
        ok, q, a = quantum(q16, w)
        rc, y, z = library_proof(shape, w, bb)
        assert rc == int(ok), (shape, w, bb)
        if not ok:
            assert not y and not z
            continue
        assert (y, z) == host_cannot_refuse(a, q, shape, bb), (shape, w, bb)
        seen.add((bb, y, z))
    assert seen == {(b, y, z) for b in (True, False) for (y, z) in ((True, True), (True, False), (False, False))}
    # the headline and the dense segmentation at 512^3, the 1024^3 volume: nothing launched behind either column pass
    assert library_proof((512, 512, 512), (6.0, 6.0, 30.0), True) == (1, True, True)
    assert library_proof((512, 512, 512), (1.0, 1.0, 1.0), False) == (1, True, True)
    assert library_proof((1024, 1024, 1024), (1.0, 1.0, 1.0), False) == (1, True, True)
    # (6, 6, 30) without a black border: pass Z's 512-row columns are beyond the 273 rows within which it carries +inf
    assert library_proof((512, 512, 512), (6.0, 6.0, 30.0), False) == (1, True, False)
    # two dimensions: pass Y only
    assert library_proof((512, 512), (1.0, 1.0), False) == (1, True, False)
