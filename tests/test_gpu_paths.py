"""GPU tests (`-m gpu`) of the kernel FAMILIES and of the Z-sharded building blocks.

1. Every column / row kernel variant against the oracle on the same inputs: the default path
   (register-resident pass 1 + wave-autonomous column pass), the workgroup-phased LDS kernels
   (debug bit 64 / 32), the fp32 form of pass 1 (bit 0x100000: no 16-bit indices between passes X and Y) and the
   size-agnostic fallback.
2. The two shard phases (edt_hip_shard_xy_device / edt_hip_shard_z_device) driven as VIRTUAL ranks
   on one device: slabs, one-slice label halo, Z-slab -> Y-slab re-partition (done with plain tensor
   slicing here; torch.distributed does it across GPUs), compared with the oracle on the whole volume.
"""
import ctypes

import numpy as np
import pytest

from synth import blocky_labels, voronoi_labels

pytestmark = pytest.mark.gpu


def same(a, b):
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


SHAPES = [(512, 40, 36), (256, 96, 20), (100, 130, 70), (64, 64, 64), (36, 500, 24), (8, 33, 257)]


@pytest.mark.parametrize("mode,name", [(0, "default"), (64, "phased-column"), (32, "lds-row"),
                                        (96, "phased-both"), (0x100000, "fp32-pass-1")])
def test_kernel_families_agree_with_the_oracle(edt_gpu, oracle_port, mode, name):
    from edt import _lib
    lib = _lib.load()
    rng = np.random.default_rng(77)
    try:
        lib.edt_hip_set_debug_mode(mode)
        for i, shape in enumerate(SHAPES):
            lab = blocky_labels(shape, nlabels=5, zero_frac=0.2 if i % 2 else 0.0,
                                block=int(rng.integers(2, 24)), rng=rng).astype(np.uint32)
            lab = np.asfortranarray(lab)
            for an, bb in (((1, 1, 1), False), ((6, 6, 30), True), ((0.5, 0.7, 1.3), False)):
                want = oracle_port.edtsq(lab, an, bb)
                got = edt_gpu.edtsq(lab, anisotropy=an, black_border=bb)
                assert same(got, want), (name, shape, an, bb)
                assert same(edt_gpu.edt(lab, anisotropy=an, black_border=bb), np.sqrt(want)), (name, shape, "sqrt")
        # a 2-D image goes through the same kernels with the fused epilogue on the first column pass
        img = np.asfortranarray(blocky_labels((300, 200), nlabels=3, zero_frac=0.1, block=9, rng=rng).astype(np.uint16))
        assert same(edt_gpu.edtsq(img, anisotropy=(2, 3), black_border=False), oracle_port.edtsq(img, (2, 3), False))
    finally:
        lib.edt_hip_set_debug_mode(0)


def test_axes_beyond_the_wave_kernels(edt_gpu, oracle_port):
    """Rows of 513..1024 voxels still take the register-resident row kernel (16 chunks per row) and axes of
    513..2048 rows the wave column kernel (2- and 1-column waves on 16-column tiles); beyond that the
    LDS-staged row kernel and the workgroup-phased column kernel take over."""
    rng = np.random.default_rng(5)
    for shape in ((1030, 40, 24), (48, 1040, 12), (40, 36, 1100), (1024, 64, 8), (600, 700, 3),
                  (20, 2048, 6), (36, 9, 1500), (17, 2049, 3), (33, 5, 2100)):
        lab = np.asfortranarray(blocky_labels(shape, nlabels=4, zero_frac=0.1, block=int(rng.integers(5, 60)),
                                              rng=rng).astype(np.uint32))
        for an, bb in (((6, 6, 30), True), ((1, 1, 1), False)):
            want = oracle_port.edtsq(lab, an, bb)
            got = edt_gpu.edtsq(lab, anisotropy=an, black_border=bb)
            assert same(got, want), (shape, an, bb)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("shape", [(96, 80, 72), (512, 64, 40), (40, 24, 33), (2200, 36, 10)])  # (the last: rows beyond the row kernels)
def test_shard_phases_as_virtual_ranks(edt_gpu, oracle_port, world, shape):
    import torch
    from edt import _lib
    from edt.distributed import HipOps, balanced_partition

    sx, sy, sz = shape
    if sz < world or sy < world:
        pytest.skip("fewer slices than ranks")
    dev = torch.device("cuda", 0)
    ops = HipOps()
    lab = voronoi_labels(shape, nseeds=40, seed=world, upsample=4, membrane=0.04)
    t = torch.from_numpy(np.ascontiguousarray(lab.T).view(np.int32)).to(dev)  # (sz, sy, sx), x fastest
    zparts, yparts = balanced_partition(sz, world), balanced_partition(sy, world)
    # (the third: voxel sizes whose multiples are not exact in fp32, the Z phase told what the XY phase guarantees --
    # edt_hip_shard_z_device_ex's field_floor, fp32 fma candidates)
    for an, bb, sqrt in (((6.0, 6.0, 30.0), True, False), ((1.0, 1.5, 0.5), False, True), ((1.1, 0.7, 1.3), False, False)):
        flags = _lib.FLAG_BLACK_BORDER if bb else 0
        partial, zflags = [], []
        for r, (zs, ze) in enumerate(zparts):
            halo = t[zs - 1].contiguous() if r > 0 else None  # the previous rank's last slice
            p, f = ops.xy(t[zs:ze].contiguous(), halo, _lib.U32, an, flags)
            partial.append(p)
            zflags.append(f)
        torch.cuda.synchronize()
        partial, zflags = torch.cat(partial, 0), torch.cat(zflags, 0)
        outs = []
        for (ys, ye) in yparts:  # the all-to-all: rank h receives (all z, its y range)
            o = ops.z(partial[:, ys:ye, :].contiguous(), zflags[:, ys:ye, :].contiguous(), an[2],
                      flags | (_lib.FLAG_SQRT if sqrt else 0), wxy=(an[0], an[1]))
            outs.append(o.clone())
        got = torch.cat(outs, 1).cpu().numpy().T
        want = oracle_port.edtsq(lab, an, bb)
        if sqrt:
            want = np.sqrt(want)
        assert same(got, want), (world, shape, an, bb)


_RECORD_CASES = {}


def _record_case(shape, oracle_port):
    """labels + oracle answers of one shape, computed once for all virtual world sizes"""
    if shape not in _RECORD_CASES:
        lab = voronoi_labels(shape, nseeds=40, seed=sum(shape), upsample=4, membrane=0.04)
        runs = [((6.0, 6.0, 30.0), True, False), ((1.0, 1.5, 0.5), False, True), ((1.1, 0.7, 1.3), True, False)]
        wants = []
        for an, bb, sqrt in runs:
            w = oracle_port.edtsq(lab, an, bb)
            wants.append(np.sqrt(w) if sqrt else w)
        _RECORD_CASES[shape] = (lab, runs, wants)
    return _RECORD_CASES[shape]


@pytest.mark.parametrize("world,chunks", [(1, 1), (2, 2), (3, 1), (8, 3)])
@pytest.mark.parametrize("shape", [(96, 280, 24), (512, 96, 20), (1024, 64, 6), (36, 1000, 9), (41, 200, 12), (24, 1500, 5), (16, 96, 1100),
                                   (2048, 96, 7), (1280, 70, 9)])
def test_shard_records_as_virtual_ranks(edt_gpu, oracle_port, world, chunks, shape):
    """The slab-record form of the two phases (edt_hip_shard_xy_records_device /
    edt_hip_shard_z_records_device): every virtual rank writes its per-destination records chunk by
    chunk (its own part straight into the receive buffer), the "exchange" is a copy of the other
    blocks, and each destination runs the Z pass over the gathered records."""
    import torch
    from edt import _lib
    from edt.distributed import HipOps, balanced_partition

    sx, sy, sz = shape
    words = -(-sy // 32)
    if sz < world or words < world:
        pytest.skip("fewer slices / y-words than ranks")
    dev = torch.device("cuda", 0)
    ops = HipOps()
    assert ops.records_supported(_lib.U32, sx, sy, sz)
    lab, runs, wants = _record_case(shape, oracle_port)
    t = torch.from_numpy(np.ascontiguousarray(lab.T).view(np.int32)).to(dev)  # (sz, sy, sx), x fastest
    zparts = balanced_partition(sz, world)
    yparts = [(32 * a, min(32 * b, sy)) for a, b in balanced_partition(words, world)]
    y_splits = [a for a, _ in yparts] + [sy]
    rec = [ops.record_floats(sx, b - a) for a, b in yparts]
    for (an, bb, sqrt), want in zip(runs, wants):
        flags = _lib.FLAG_BLACK_BORDER if bb else 0
        dst = [torch.full((sz, rec[h]), float("nan"), dtype=torch.float32, device=dev) for h in range(world)]
        for r, (zs, ze) in enumerate(zparts):
            halo = t[zs - 1] if r > 0 else None  # the previous rank's last slice
            for c0, c1 in balanced_partition(ze - zs, min(chunks, ze - zs)):
                blocks = [dst[h][zs + c0:zs + c1] if h == r else
                          torch.empty((c1 - c0, rec[h]), dtype=torch.float32, device=dev) for h in range(world)]
                ops.xy_records(t[zs + c0:zs + c1], halo, _lib.U32, an, flags, y_splits, blocks)
                for h in range(world):
                    if h != r:
                        dst[h][zs + c0:zs + c1].copy_(blocks[h])  # the exchange
                halo = t[zs + c1 - 1]
        outs = []
        for h, (ys, ye) in enumerate(yparts):
            # (the Z phase is told the voxel sizes of the XY phase: integer column kernel / fp32 fma candidates, edt_hip.h;
            # every other rank without them -- the fp32 kernels with fp64 candidates: the same bits)
            ops.z_records(dst[h], sx, ye - ys, an[2], flags | (_lib.FLAG_SQRT if sqrt else 0),
                          wxy=(an[0], an[1]) if h % 2 == 0 else None)
            outs.append(dst[h][:, :(ye - ys) * sx].reshape(sz, ye - ys, sx))
        got = torch.cat(outs, 1).cpu().numpy().T
        assert same(got, want), (world, shape, an, bb)


@pytest.mark.parametrize("world,chunks", [(1, 1), (2, 2), (3, 1), (8, 3)])
@pytest.mark.parametrize("shape", [(96, 280, 100), (512, 128, 97), (64, 1000, 130), (36, 200, 1024), (1024, 256, 104)])
def test_shard_records16_as_virtual_ranks(edt_gpu, oracle_port, world, chunks, shape):
    """Slab records of 16-bit rows (edt_hip_shard_xy_records16_device / edt_hip_shard_z_records16_device: 2.25 bytes per
    voxel): the same virtual ranks, voxel sizes that share a quantum, both scan axes on the integer kernel (97..1024 rows).
    No tile is refused on these labels (the counter stays 0) and the gathered result is the oracle's, bit for bit."""
    import torch
    from edt import _lib
    from edt.distributed import HipOps, balanced_partition

    sx, sy, sz = shape
    words = -(-sy // 32)
    if sz < world or words < world:
        pytest.skip("fewer slices / y-words than ranks")
    dev = torch.device("cuda", 0)
    ops = HipOps()
    lab = voronoi_labels(shape, nseeds=60, seed=sum(shape), upsample=4, membrane=0.04)
    t = torch.from_numpy(np.ascontiguousarray(lab.T).view(np.int32)).to(dev)
    zparts = balanced_partition(sz, world)
    yparts = [(32 * a, min(32 * b, sy)) for a, b in balanced_partition(words, world)]
    y_splits = [a for a, _ in yparts] + [sy]
    rec = [ops.record16_words(sx, b - a) for a, b in yparts]
    assert rec[0] < ops.record_floats(sx, yparts[0][1] - yparts[0][0]) * 0.55
    for an, bb, sqrt in (((6.0, 6.0, 30.0), True, False), ((1.0, 1.0, 1.0), True, True), ((2.0, 1.0, 3.0), True, False)):
        assert ops.records16_supported(_lib.U32, sx, sy, sz, an)
        flags = _lib.FLAG_BLACK_BORDER if bb else 0
        refused = torch.zeros(1, dtype=torch.int32, device=dev)
        dst = [torch.full((sz, rec[h]), -1, dtype=torch.int32, device=dev) for h in range(world)]
        for r, (zs, ze) in enumerate(zparts):
            halo = t[zs - 1] if r > 0 else None
            for c0, c1 in balanced_partition(ze - zs, min(chunks, ze - zs)):
                blocks = [dst[h][zs + c0:zs + c1] if h == r else
                          torch.empty((c1 - c0, rec[h]), dtype=torch.int32, device=dev) for h in range(world)]
                ops.xy_records16(t[zs + c0:zs + c1], halo, _lib.U32, an, flags, y_splits, blocks, refused)
                for h in range(world):
                    if h != r:
                        dst[h][zs + c0:zs + c1].copy_(blocks[h])  # the exchange
                halo = t[zs + c1 - 1]
        assert int(refused.item()) == 0
        outs = []
        for h, (ys, ye) in enumerate(yparts):
            out = torch.full((sz, ye - ys, sx), float("nan"), dtype=torch.float32, device=dev)
            ops.z_records16(dst[h], out, an, flags | (_lib.FLAG_SQRT if sqrt else 0))
            outs.append(out)
        got = torch.cat(outs, 1).cpu().numpy().T
        want = oracle_port.edtsq(lab, an, bb)
        assert same(got, np.sqrt(want) if sqrt else want), (world, shape, an, bb)


def test_shard_records16_count_the_tiles_without_a_16_bit_form(edt_gpu, oracle_port):
    """a label 300 voxels across at (1, 1, 1): k^2 leaves 16 bits from k = 256 on -- the XY phase counts those tiles (and
    nothing else is promised about them); rows without a boundary (no black border) likewise; a shallow volume counts none.
    And the Z phase serves values beyond ITS limit: (1, 1, 6) -- a_z = 36, limit 36 * 42^2 < the Y pass's -- through fp32."""
    import torch
    from edt import _lib
    from edt.distributed import HipOps

    dev = torch.device("cuda", 0)
    ops = HipOps()
    sx, sy, sz = 640, 128, 100
    lab = np.ones((sx, sy, sz), dtype=np.uint32, order="F")
    lab[:, :, :50] = voronoi_labels((sx, sy, 50), nseeds=80, seed=5, upsample=4, membrane=0.04)
    t = torch.from_numpy(np.ascontiguousarray(lab.T).view(np.int32)).to(dev)
    rec = ops.record16_words(sx, sy)
    for bb, lo, hi in ((True, 0, 50), (True, 50, 100), (False, 50, 100)):
        refused = torch.zeros(1, dtype=torch.int32, device=dev)
        blk = torch.empty((hi - lo, rec), dtype=torch.int32, device=dev)
        ops.xy_records16(t[lo:hi], t[lo - 1] if lo else None, _lib.U32, (1.0, 1.0, 1.0), _lib.FLAG_BLACK_BORDER if bb else 0,
                         [0, sy], [blk], refused)
        n = int(refused.item())
        assert (n == 0) if hi == 50 else (n > 0), (bb, lo, hi, n)
    # the Z phase's own limit: one rank, one box 508 voxels across -- k <= 254, N <= 64 516 fits the XY phase (limit 255^2), not
    # the Z pass at a_z = 36 (limit 36 * 42^2 = 63 504): those tiles get their rows as fp32 values and go to the fp32 kernel
    shape = (508, 508, 100)
    lab = np.ones(shape, dtype=np.uint32, order="F")
    lab[100:104, 37, :] = 0
    t = torch.from_numpy(np.ascontiguousarray(lab.T).view(np.int32)).to(dev)
    an = (1.0, 1.0, 6.0)
    refused = torch.zeros(1, dtype=torch.int32, device=dev)
    blk = torch.empty((shape[2], ops.record16_words(shape[0], shape[1])), dtype=torch.int32, device=dev)
    ops.xy_records16(t, None, _lib.U32, an, _lib.FLAG_BLACK_BORDER, [0, shape[1]], [blk], refused)
    assert int(refused.item()) == 0
    out = torch.empty((shape[2], shape[1], shape[0]), dtype=torch.float32, device=dev)
    ops.z_records16(blk, out, an, _lib.FLAG_BLACK_BORDER)
    assert same(out.cpu().numpy().T, oracle_port.edtsq(lab, an, True))


def test_device_entry_point_is_graph_capturable(edt_gpu, oracle_port):
    """edt_hip_edtsq_device only enqueues kernels on the caller's stream (no allocation, no synchronisation):
    a whole transform can be captured into a hipGraph and replayed (launch-bound small volumes)."""
    import torch
    from edt import device

    dev = torch.device("cuda", 0)
    lab = voronoi_labels((96, 80, 72), nseeds=40, seed=4, upsample=4, membrane=0.04)
    t = torch.from_numpy(np.ascontiguousarray(lab.T).view(np.int32)).to(dev)
    out = torch.empty(t.shape, dtype=torch.float32, device=dev)
    plan = device.Plan(lab.shape, 2, dev)
    an = (6.0, 6.0, 30.0)
    plan.run(t, an, black_border=False, out=out)  # warm-up: lazy code-object loads, function attributes
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            plan.run(t, an, black_border=False, out=out)
    want = oracle_port.edtsq(lab, an, False)
    for _ in range(3):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert same(out.cpu().numpy().T, want)


def test_short_axes_thread_per_column(edt_gpu, oracle_port):
    """Axes of at most 32 rows with many columns take edt_short.hip (a thread per column, rows in registers, brute
    force over the column) instead of single-wave workgroups of the LDS-tiled kernel; both against the oracle, and
    against each other (debug bit 0x1000000 keeps the wave kernel)."""
    from edt import _lib
    lib = _lib.load()
    rng = np.random.default_rng(31)
    shapes = [(300, 200, 8), (257, 130, 17), (130, 5, 400), (66, 32, 70), (512, 512, 3), (100, 100, 1), (90, 31, 33),
              (4100, 7, 9)]
    for t, shape in enumerate(shapes):
        dt = [np.uint8, np.uint16, np.uint32, np.uint64, np.float32][t % 5]
        lab = np.asfortranarray(blocky_labels(shape, nlabels=4, zero_frac=0.2, block=int(rng.integers(1, 7)), rng=rng).astype(dt))
        if t % 3 == 0:
            lab[:, :, 0] = 1   # whole slices of one label: runs that span the short axis, no boundary in some rows
        for an, bb in (((1.0, 1.0, 1.0), False), ((6.0, 6.0, 30.0), True), ((0.7, 1.3, 2.1), False)):
            want = oracle_port.edtsq(lab, an, bb)
            got = edt_gpu.edtsq(lab, anisotropy=an, black_border=bb)
            assert same(got, want), (shape, dt, an, bb)
            assert same(edt_gpu.edt(lab, anisotropy=an, black_border=bb), np.sqrt(want)), (shape, "sqrt")
            lib.edt_hip_set_debug_mode(0x1000000)
            try:
                assert same(edt_gpu.edtsq(lab, anisotropy=an, black_border=bb), want), (shape, "wave kernel")
            finally:
                lib.edt_hip_set_debug_mode(0)
    # the binary route and a 2-D image with a short y axis
    img = np.asfortranarray(blocky_labels((5000, 20), nlabels=3, zero_frac=0.3, block=3, rng=rng).astype(np.uint8))
    assert same(edt_gpu.edtsq(img, anisotropy=(2.0, 3.0), black_border=True), oracle_port.edtsq(img, (2.0, 3.0), True))
    assert same(edt_gpu.binary_edtsq(img, anisotropy=(2.0, 3.0), black_border=False), oracle_port.binary_edtsq(img, (2.0, 3.0), False))


@pytest.mark.parametrize("sx", [1025, 1088, 1280, 1281, 1535, 1536, 1537, 1600, 1792, 1800, 2047, 2048,
                                2049, 2304, 2560, 3000, 3071, 3072, 3073, 3500, 4095, 4096])
def test_rows_of_1025_to_2048_voxels_two_waves_per_row(edt_gpu, oracle_port, sx):
    """Rows of 1025..2048 (4096) voxels: pass X as two (four) waves per row (edt_rowwave.hip, H = 2 / 4) that exchange
    positions through LDS -- every part the last and the first run start of its own voxels.  Label patterns that stress
    the exchange: runs that cross the parts' boundaries, rows whose starts all lie in ONE part, rows without any start, a
    start exactly at the first / last voxel of a part, single-voxel runs; odd numbers of rows in the last y-band (the
    exchange words alternate from row to row across groups).  Against the oracle, with the 16-bit index form and with
    fp32 between X and Y, and against the other kernels of pass X (debug bit 0x4000000: the workgroup-phased kernel up
    to 2048 voxels, the line pipeline beyond)."""
    from edt import _lib
    lib = _lib.load()
    rng = np.random.default_rng(sx)
    nc = -(-sx // 64)
    # voxels per wave (edt_rowwave.hip: launch_row_wave_t)
    half = 64 * ((10 if nc <= 20 else 12 if nc <= 24 else 14 if nc <= 28 else 16) if nc <= 32 else (12 if nc <= 48 else 16))
    sy, sz = (37, 3) if sx % 2 else (70, 2)
    dt = [np.uint8, np.uint16, np.uint32, np.uint64, np.float32][sx % 5]
    lab = blocky_labels((sx, sy, sz), nlabels=3, zero_frac=0.15, block=int(rng.integers(200, 500)), rng=rng)
    lab[:, 0, :] = 1                              # rows without any run start
    lab[:half, 1, :] = 1; lab[half:, 1, :] = 2    # one start, exactly at the first voxel of the right half
    lab[:half - 1, 2, :] = 1; lab[half - 1:, 2, :] = 2  # ... at the last voxel of the left half
    lab[:, 3, :] = 1; lab[5, 3, :] = 0            # starts in the left half only
    lab[:, 4, :] = 2; lab[sx - 3, 4, :] = 1       # ... in the right half only
    lab[:, 5, :] = rng.integers(0, 3, size=(sx, sz))  # single-voxel runs everywhere
    lab[:, 6, :] = 0                              # background rows
    for k in (2, 3):                              # (four waves per row) starts at the later parts' first / last voxels
        if k * half < sx:
            lab[:k * half, 5 + 2 * k, :] = 1; lab[k * half:, 5 + 2 * k, :] = 2
            lab[:k * half - 1, 6 + 2 * k, :] = 3; lab[k * half - 1:, 6 + 2 * k, :] = 1
    lab[:, 13, :] = 1; lab[sx // 2, 13, :] = 2     # one single-voxel run in the middle: every part looks past its neighbours
    lab = np.asfortranarray(lab.astype(dt))
    for an, bb in (((1.0, 1.0, 1.0), False), ((6.0, 6.0, 30.0), True), ((0.7, 1.3, 2.1), False), ((1.0, 2.0, 1.0), True)):
        want = oracle_port.edtsq(lab, an, bb)
        for mode in (0, 0x100000, 0x4000000):
            lib.edt_hip_set_debug_mode(mode)
            try:
                got = edt_gpu.edtsq(lab, anisotropy=an, black_border=bb)
            finally:
                lib.edt_hip_set_debug_mode(0)
            assert same(got, want), (sx, dt, an, bb, hex(mode), int((got != want).sum()))
    img = np.asfortranarray(lab[:, :, 0])         # 2-D (no z bits)
    assert same(edt_gpu.edtsq(img, anisotropy=(1.0, 1.0), black_border=False), oracle_port.edtsq(img, (1.0, 1.0), False))
