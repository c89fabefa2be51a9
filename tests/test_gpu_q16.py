"""GPU tier: the 16-bit integer column kernel (csrc/edt_colq16.hip) and its hand-over to the fp32 kernel.

Every case is compared bit for bit with the oracle, and with this library's own fp32 kernels (debug bit 0x8000000: no
integer kernel; 0x10000000: integer kernels with fp32 values between passes Y and Z instead of the 16-bit plane).  The
shapes are chosen so that tiles are refused for every reason there is: rows without any boundary (black_border off), objects
deeper than 255 voxels (values beyond 16 bits), axes of more than 512 rows (the fp32 kernel then works on 16-column tiles:
two list entries per refused tile), volumes of several index slabs (no plane), and partial tiles."""
import numpy as np
import pytest

from synth import blocky_labels, voronoi_labels

pytestmark = pytest.mark.gpu

MODES = [(0, "q16 + plane"), (0x10000000, "q16, fp32 between Y and Z"), (0x8000000, "fp32 kernels"),
         (0x40000000, "q16, tiles beyond 16 bits as two wide passes (no column subsets)"),
         (0x20000000, "q16 without the wide form: tiles beyond 16 bits on the fp32 kernel, every list launched")]


def run_modes(edt_gpu, lab, an, bb):
    from edt import _lib
    lib = _lib.load()
    outs = []
    try:
        for mode, _ in MODES:
            lib.edt_hip_set_debug_mode(mode)
            outs.append(edt_gpu.edtsq(lab, anisotropy=an, black_border=bb))
    finally:
        lib.edt_hip_set_debug_mode(0)
    return outs


@pytest.mark.parametrize("shape", [(160, 300, 140), (40, 1000, 36), (72, 136, 1024), (512, 512, 20), (260, 200, 130),
                                   (100, 640, 520), (36, 128, 128)])
def test_q16_against_oracle_and_fp32_kernels(edt_gpu, oracle_port, shape):
    rng = np.random.default_rng(sum(shape))
    labs = [voronoi_labels(shape, nseeds=30, seed=sum(shape), upsample=4, membrane=0.03),
            blocky_labels(shape, nlabels=3, zero_frac=0.3, block=int(rng.integers(20, 90)), rng=rng).astype(np.uint16)]
    for lab in labs:
        lab = np.asfortranarray(lab)
        for an, bb in (((1, 1, 1), False), ((6, 6, 30), True), ((1.0, 1.5, 0.5), False), ((4, 4, 40), True)):
            want = oracle_port.edtsq(lab, an, bb)
            for got, (_, name) in zip(run_modes(edt_gpu, lab, an, bb), MODES):
                assert np.array_equal(got, want), (shape, an, bb, name)
            assert np.array_equal(edt_gpu.edt(lab, anisotropy=an, black_border=bb), np.sqrt(want)), (shape, an, bb, "sqrt")


def test_q16_deep_objects_leave_the_16_bit_range(edt_gpu, oracle_port):
    """a box 600 voxels deep: the middle of its rows holds x-distances of up to 300 voxels -- more than 16 bits of squares;
    those tiles go to the fp32 kernel, their neighbours stay on the integer kernel (closed form: the box)"""
    from synth import box_edtsq_closed_form
    lab = np.ones((600, 300, 280), dtype=np.uint8, order="F")
    for an in ((1.0, 1.0, 1.0), (6.0, 6.0, 30.0)):
        want = box_edtsq_closed_form(lab.shape, an)
        for got, (_, name) in zip(run_modes(edt_gpu, lab, an, True), MODES):
            assert np.array_equal(got, want), (an, name)
    # the same object inside a background: borders from the labels, black_border off
    lab2 = np.zeros((640, 260, 200), dtype=np.uint8, order="F")
    lab2[20:620, 10:250, 8:192] = 1
    want = oracle_port.edtsq(lab2, (1, 1, 1), False)
    for got, (_, name) in zip(run_modes(edt_gpu, lab2, (1, 1, 1), False), MODES):
        assert np.array_equal(got, want), name


WIDE_SHAPES = [(640, 130, 140), (1100, 130, 100), (560, 520, 130), (2048, 130, 132), (600, 1024, 36), (640, 40, 1024),
               (532, 260, 100), (528, 130, 100)]


@pytest.mark.parametrize("shape", WIDE_SHAPES)
def test_q16_wide_form(edt_gpu, oracle_port, shape):
    """Tiles holding values beyond 16 bits -- the middle of rows of more than 510 voxels, objects deeper than ~255 voxels -- are
    worked on by the integer kernel itself as two half-tiles of 16 columns with 32-bit lanes (csrc/edt_colq16.hip, go_wide):
    against the oracle; against the same library without the wide form (0x20000000), with fp32 values between the passes
    (0x10000000), with the fp32 form of pass X (0x100000: the Y pass reads fp32 values) and on the fp32 kernels.  Shapes: row
    lengths that end in a half tile or a quarter of one, 1024-row axes (workgroups of 512 threads), deep multi-label cells."""
    from edt import _lib
    lib = _lib.load()
    rng = np.random.default_rng(sum(shape))
    labs = [np.ones(shape, dtype=np.uint32, order="F"),
            np.asfortranarray(blocky_labels(shape, nlabels=3, zero_frac=0.0, block=int(rng.integers(270, 420)), rng=rng).astype(np.uint32))]
    labs[1][rng.random(shape) < 0.0002] = 0
    combos = (((1, 1, 1), True), ((6, 6, 30), True), ((1, 1, 1), False), ((30, 6, 6), True), ((0.5, 0.5, 1.0), True))
    if shape[0] * shape[1] * shape[2] > 15_000_000:
        combos = combos[1:3]  # (the oracle's time on the larger volumes)
    for lab in labs:
        for an, bb in combos:
            want = oracle_port.edtsq(lab, an, bb)
            try:
                # (0x40000000: every tile beyond 16 bits as two wide passes; default: its marked columns alone where there are <= 16)
                for mode in (0, 0x40000000, 0x20000000, 0x10000000, 0x100000, 0x100000 | 0x40000000, 0x8000000):
                    lib.edt_hip_set_debug_mode(mode)
                    got = edt_gpu.edtsq(lab, anisotropy=an, black_border=bb)
                    assert np.array_equal(got, want), (shape, an, bb, hex(mode))
            finally:
                lib.edt_hip_set_debug_mode(0)
            assert np.array_equal(edt_gpu.edt(lab, anisotropy=an, black_border=bb), np.sqrt(want)), (shape, an, bb, "sqrt")


def slab_labels(shape, rng):
    """one label from edge to edge in most rows (no boundary along x: +inf after pass X without a black border), thin slabs of
    other labels at the low end of y and of z (borders far from most rows), a few background voxels (finite sites)"""
    sx, sy, sz = shape
    lab = np.ones(shape, dtype=np.uint32, order="F")
    lab[:, :3, :] = 2
    lab[: sx // 2, :, :2] = 3
    for _ in range(4):
        lab[rng.integers(0, sx), rng.integers(0, sy), rng.integers(0, sz)] = 0
    return lab


@pytest.mark.parametrize("shape", [(16, 413, 216), (72, 518, 352), (32, 300, 300), (64, 140, 260)])
def test_q16_rows_without_boundary_and_far_borders(edt_gpu, oracle_port, shape):
    """black_border off, rows without any boundary: +inf in the integer kernel's wide form (round 5) -- carried only where the
    column is short enough for the border distance of such a row (anything up to the column's length) and for a finite
    neighbour's value plus a * d^2 to stay exact (csrc/edt_colq16_lane.h: q16_wide_range); handed to the fp32 kernel elsewhere;
    and the host's proof that no list launch is needed has to allow for what pass Y makes of +inf.  The shapes and voxel sizes
    are those of the GPU fuzz's mismatches on the first version ((0.5, 40, 2) / (6, 40, 3) on columns of 413 / 518 rows) and
    of the CPU tier's separating case ((1, 10, 10) on 300 x 300)."""
    rng = np.random.default_rng(sum(shape))
    labs = [slab_labels(shape, rng),
            np.asfortranarray(blocky_labels(shape, nlabels=4, zero_frac=0.05, block=int(rng.integers(20, 120)), rng=rng).astype(np.uint32))]
    for lab in labs:
        for an in ((0.5, 40.0, 2.0), (6.0, 40.0, 3.0), (1.0, 10.0, 10.0), (1.0, 1.0, 1.0), (30.0, 6.0, 2.0)):
            want = oracle_port.edtsq(lab, an, False)
            for got, (_, name) in zip(run_modes(edt_gpu, lab, an, False), MODES):
                assert np.array_equal(got, want), (shape, an, name)
        lab_c = np.ascontiguousarray(lab[:, :, : shape[2] // 2])
        want = oracle_port.edtsq(lab_c, (2.0, 40.0, 0.5), False)
        assert np.array_equal(edt_gpu.edtsq(lab_c, anisotropy=(2.0, 40.0, 0.5), black_border=False), want), (shape, "C order")
        assert np.array_equal(edt_gpu.edt(lab, anisotropy=(1.0, 10.0, 10.0), black_border=False),
                              np.sqrt(oracle_port.edtsq(lab, (1.0, 10.0, 10.0), False))), (shape, "sqrt")


@pytest.mark.parametrize("shape", [(128, 300, 200), (96, 512, 130), (64, 130, 1000), (36, 100, 100), (128, 128, 128)])
def test_q16_tiles_of_nothing_but_inf(edt_gpu, oracle_port, shape):
    """Round 6: without a black border a tile inside one object that spans the volume along the earlier axes holds nothing but
    +inf and no run start -- it is answered from the fill (+inf row for row; in pass Y it may stay in the 16-bit plane as the 0xFFFF
    pass X left there) instead of going through the 32-bit form.  Volumes made of such tiles, of tiles that just are not (one
    finite voxel, one label change behind row 0), and of both: the oracle's results under the default form selection, without
    the short cut (0x80), with fp32 between the passes (0x10000000) and on the fp32 kernels (0x8000000); edt and edtsq."""
    from edt import _lib
    lib = _lib.load()
    sx, sy, sz = shape
    vols = []
    ones = np.ones(shape, dtype=np.uint32, order="F")
    vols.append(("ones", ones))
    v = ones.copy(order="F"); v[sx // 2, sy // 2, sz // 2] = 0
    vols.append(("one background voxel", v))
    v = ones.copy(order="F"); v[:, : sy // 3, :] = 2                      # run starts along y in every column, nothing along x
    vols.append(("two slabs along y", v))
    v = ones.copy(order="F"); v[:, :, sz // 2:] = 3                      # run starts along z only
    vols.append(("two slabs along z", v))
    v = ones.copy(order="F"); v[: sx // 2, : sy // 2, : sz // 2] = 4     # a corner: finite rows in an eighth of the volume
    vols.append(("a corner block", v))
    for name, lab in vols:
        for an in ((1.0, 1.0, 1.0), (6.0, 6.0, 30.0), (0.5, 1.0, 2.0)):
            want = oracle_port.edtsq(lab, an, False)
            try:
                for mode in (0, 0x80, 0x10000000, 0x8000000):
                    lib.edt_hip_set_debug_mode(mode)
                    assert np.array_equal(edt_gpu.edtsq(lab, anisotropy=an, black_border=False), want), (shape, name, an, hex(mode))
                    if mode in (0, 0x80):
                        assert np.array_equal(edt_gpu.edt(lab, anisotropy=an, black_border=False), np.sqrt(want)), (shape, name, an, hex(mode), "sqrt")
            finally:
                lib.edt_hip_set_debug_mode(0)
    # two dimensions: pass Y is the last pass
    img = np.ones((sx, sy), dtype=np.uint8, order="F")
    assert np.array_equal(edt_gpu.edtsq(img, black_border=False), oracle_port.edtsq(img, (1.0, 1.0), False))
    img[:, sy // 2:] = 2
    assert np.array_equal(edt_gpu.edtsq(img, black_border=False), oracle_port.edtsq(img, (1.0, 1.0), False))


@pytest.mark.parametrize("shape", [(128, 200, 160), (100, 512, 130), (64, 130, 1000), (36, 97, 100), (512, 128, 128), (640, 130, 140), (528, 100, 600)])
def test_q16_tiles_without_structure(edt_gpu, oracle_port, shape):
    """Round 6: a tile with no run start behind row 0 and every row equal to row 0 (the inside of a box) is answered from its
    image -- min(N, the border parabola of the column's ends), or N without a black border -- without scans, break bits or
    blocks.  Volumes made of such tiles and of tiles that JUST are not: one voxel of another label (a run start in one column, in
    y and in z), one row whose x-profile differs (no run start along the scan axis, but rows that are not equal), partial tiles
    (row lengths that are no multiple of 32), columns that end inside a band.  The oracle's results under the default selection,
    without the short cut (0x80), with fp32 between the passes (0x10000000) and on the fp32 kernels (0x8000000); edt and edtsq;
    both border rules; voxel sizes whose border parabolas leave 16 bits early (a = 25, 1600)."""
    from edt import _lib
    lib = _lib.load()
    sx, sy, sz = shape
    ones = np.ones(shape, dtype=np.uint16, order="F")
    vols = [("one label", ones)]
    v = ones.copy(order="F"); v[sx // 3, sy // 2, sz // 2] = 2
    vols.append(("one voxel of another label", v))
    v = ones.copy(order="F"); v[: sx // 2, sy // 4, :] = 3                 # one y-row of every slice with another x-profile ...
    v[: sx // 2, sy // 4 + 1:, :] = 3                                       # ... continued upwards: rows differ, run starts along y too
    vols.append(("a step along y", v))
    v = ones.copy(order="F"); v[sx - 5:, :, :] = 0                           # background at the end of every row: flat along y and z
    vols.append(("background slab along x", v))
    v = ones.copy(order="F"); v[:, :, : sz // 3] = 4
    vols.append(("two slabs along z", v))
    for name, lab in vols:
        for an in ((1.0, 1.0, 1.0), (6.0, 6.0, 30.0), (1.0, 40.0, 5.0)):
            for bb in (True, False):
                want = oracle_port.edtsq(lab, an, bb)
                try:
                    for mode in (0, 0x80, 0x10000000, 0x8000000):
                        lib.edt_hip_set_debug_mode(mode)
                        assert np.array_equal(edt_gpu.edtsq(lab, anisotropy=an, black_border=bb), want), (shape, name, an, bb, hex(mode))
                        if mode == 0:
                            assert np.array_equal(edt_gpu.edt(lab, anisotropy=an, black_border=bb), np.sqrt(want)), (shape, name, an, bb, "sqrt")
                finally:
                    lib.edt_hip_set_debug_mode(0)
    img = np.ones((sx, sy), dtype=np.uint8, order="F")                      # two dimensions: pass Y is the last pass
    for bb in (True, False):
        assert np.array_equal(edt_gpu.edt(img, anisotropy=(2.0, 3.0), black_border=bb), np.sqrt(oracle_port.edtsq(img, (2.0, 3.0), bb)))


def test_q16_refused_tile_with_inf_rows_in_the_plane(edt_gpu, oracle_port):
    """Found by the round-6 fuzz on the first build with the short cut above (1 of 600 cases): slices of nothing but +inf that pass Y
    left in the 16-bit plane (0xFFFF) next to a slice whose values pass Z's integer form cannot hold (x-distances of up to 211
    voxels at a voxel size of 40: N = k^2 * 1600 beyond 2^24) -- pass Z REFUSES such a tile and first turns its plane rows into fp32
    values for the fp32 kernel: 0xFFFF must become FLT_MAX there, not 65535 quanta."""
    shape = (212, 189, 253)
    lab = np.ones(shape, dtype=bool, order="F")
    lab[0, :, 10] = False                       # one slice with a boundary at x = 0: indices up to 211 along x
    lab[5:9, 100:140, 200:203] = False          # and some ordinary structure elsewhere
    for an in ((40.0, 2.0, 3.0), (40.0, 40.0, 2.0)):
        want = oracle_port.edtsq(lab, an, False)
        assert np.isfinite(want).all() and want.max() > 2 ** 24
        got = edt_gpu.edtsq(lab, anisotropy=an, black_border=False)
        assert np.array_equal(got, want), (an, int((got != want).sum()))


def test_q16_two_dimensional_and_stacks(edt_gpu, oracle_port):
    rng = np.random.default_rng(3)
    for shape in ((300, 260), (1000, 200), (128, 1024)):
        img = np.asfortranarray(blocky_labels(shape, nlabels=6, zero_frac=0.1, block=17, rng=rng).astype(np.uint32))
        for an, bb in (((1, 1), True), ((2, 3), False), ((6, 30), True)):
            want = oracle_port.edtsq(img, an, bb)
            for got, (_, name) in zip(run_modes(edt_gpu, img, an, bb), MODES):
                assert np.array_equal(got, want), (shape, an, bb, name)
            assert np.array_equal(edt_gpu.edt(img, anisotropy=an, black_border=bb), np.sqrt(want))


def test_q16_binary_route_and_bool(edt_gpu, oracle_port):
    rng = np.random.default_rng(9)
    lab = np.asfortranarray((rng.random((200, 180, 150)) < 0.97))
    want = oracle_port.edtsq(lab, (1, 1, 2), True)
    for got, (_, name) in zip(run_modes(edt_gpu, lab, (1, 1, 2), True), MODES):
        assert np.array_equal(got, want), name


@pytest.mark.parametrize("shape", [(64, 80, 72), (128, 200, 100), (36, 300, 70), (66, 50, 40), (260, 260)])
def test_q16_voxel_graph_output_stride_two(edt_gpu, oracle_port, shape):
    """the doubled grids of the voxel-graph transform on the integer kernel (blocks of 16 rows, even rows evaluated, the last
    pass writing the caller's array): against the oracle and against the fp32 kernels; band counts odd and even.  Rows of
    whole granules take the index form (16-bit indices out of pass X, compact even rows out of pass Y): debug bit 0x100000
    keeps the fp32 form, 0x8000000 the fp32 kernels on either form."""
    from synth import blob_mask
    from edt import _lib
    lib = _lib.load()
    rng = np.random.default_rng(sum(shape))
    lab = (blob_mask(shape, rng=rng, p=0.8, block=7) * rng.integers(1, 4, size=shape)).astype(np.uint8)
    g = np.full(shape, 0b00111111, dtype=np.uint8)
    for bit in (0x01, 0x04, 0x10):
        g[rng.random(shape) < 0.03] &= np.uint8(~bit & 0xFF)
    for an, bb in (((6.0, 6.0, 30.0), True), ((1.0, 1.0, 1.0), False), ((1.0, 2.0, 1.5), True)):
        an = an[:len(shape)]
        want = oracle_port.edtsq(lab, an, bb, voxel_graph=g)
        try:
            for mode in (0, 0x8000000, 0x100000, 0x100000 | 0x8000000):
                lib.edt_hip_set_debug_mode(mode)
                got = edt_gpu.edtsq(lab, anisotropy=an, black_border=bb, voxel_graph=g)
                assert np.array_equal(got, want, equal_nan=True), (shape, an, bb, hex(mode))
        finally:
            lib.edt_hip_set_debug_mode(0)
        assert np.array_equal(edt_gpu.edt(lab, anisotropy=an, black_border=bb, voxel_graph=g), np.sqrt(want), equal_nan=True)


@pytest.mark.parametrize("pad", [8, 4096, 8192 + 24])
def test_q16_padded_index_buffer(edt_gpu, oracle_port, pad, monkeypatch):
    """Round 6: the index buffer of pass X -- which becomes the 16-bit plane between passes Y and Z -- may keep its slices further
    apart than sx * sy elements (csrc/edt_api.hip: plane_pad_elems; slices that are a multiple of 1 MiB alias in pass Z, the
    full-size shapes of tests/test_gpu_fullsize.py get their pad from that rule).  Forced here (EDT_HIP_PLANE_PAD_BYTES, read per
    plan) onto small volumes whose tiles take every way through the passes: qualifying tiles (pass Y writes the plane over the
    indices, pass Z reads it at its own row stride), tiles beyond 16 bits (the wide form re-reads the indices), refused tiles (the
    fp32 kernel reads the indices of its tile through XFuse.c_outer), tiles of nothing but +inf and tiles without structure (both
    short cuts leave / write plane rows), partial tiles; the signed transform (the sign in pass Z's epilogue); under the form bits
    of MODES; edt and edtsq.  A pad of 8 bytes separates every stride that should be the pitch from sx * sy."""
    monkeypatch.setenv("EDT_HIP_PLANE_PAD_BYTES", str(pad))
    from edt import _lib
    lib = _lib.load()
    rng = np.random.default_rng(pad)
    # (the plan of a 3-D call really takes the pad: sz slices of 16-bit elements more workspace than without)
    with_pad = lib.edt_hip_workspace_bytes_flags(0, 3, 160, 300, 140, 0)
    monkeypatch.setenv("EDT_HIP_PLANE_PAD_BYTES", "0")
    assert with_pad - lib.edt_hip_workspace_bytes_flags(0, 3, 160, 300, 140, 0) >= 140 * (pad & ~7) - 4096
    monkeypatch.setenv("EDT_HIP_PLANE_PAD_BYTES", str(pad))
    for shape in ((160, 300, 140), (72, 136, 1024), (640, 130, 140), (36, 97, 100), (512, 64, 33)):
        labs = [np.asfortranarray(voronoi_labels(shape, nseeds=30, seed=sum(shape), upsample=4, membrane=0.03)),
                np.asfortranarray(blocky_labels(shape, nlabels=3, zero_frac=0.3, block=int(rng.integers(20, 300)), rng=rng).astype(np.uint16)),
                np.ones(shape, dtype=np.uint8, order="F")]
        for lab in labs:
            for an, bb in (((1, 1, 1), False), ((6, 6, 30), True), ((1.0, 1.5, 0.5), False)):
                want = oracle_port.edtsq(lab, an, bb)
                for got, (_, name) in zip(run_modes(edt_gpu, lab, an, bb), MODES):
                    assert np.array_equal(got, want), (pad, shape, an, bb, name)
                for mode in (0x80, 0x400):
                    try:
                        lib.edt_hip_set_debug_mode(mode)
                        assert np.array_equal(edt_gpu.edtsq(lab, anisotropy=an, black_border=bb), want), (pad, shape, an, bb, hex(mode))
                    finally:
                        lib.edt_hip_set_debug_mode(0)
                assert np.array_equal(edt_gpu.edt(lab, anisotropy=an, black_border=bb), np.sqrt(want)), (pad, shape, an, bb, "sqrt")
        lab = labs[1]
        want = oracle_port.edtsq(lab, (6, 6, 30), True) - oracle_port.edtsq(lab == 0, (6, 6, 30), True)
        assert np.array_equal(edt_gpu.sdfsq(lab, anisotropy=(6, 6, 30), black_border=True), want), (pad, shape, "sdfsq")
