"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP path, called through the C ABI
via the drop-in Python module, against the CPU oracle on the same seeded inputs.

Bar: squared distances bit-identical (`np.array_equal`); `edt` (with the fused correctly
rounded sqrt) bit-identical as well, i.e. 0 ULP (the north-star allows 1 ULP)."""
import numpy as np
import pytest

from conftest import load_golden
from synth import blocky_labels, blob_mask, box_edtsq_closed_form, config_volume, voronoi_labels

pytestmark = pytest.mark.gpu

DTYPES = [np.uint8, np.uint16, np.uint32, np.uint64, np.int8, np.int32, np.int64, np.float32,
          np.float64, bool]
ANISO = [(1, 1, 1), (6, 6, 30), (4, 4, 40), (0.5, 0.7, 1.3), (3, 1, 2), (1e-3, 2.5, 7)]


def same(a, b):
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b, equal_nan=True)


def explain(got, want):
    bad = np.argwhere(~((got == want) | (np.isnan(got) & np.isnan(want))))
    if len(bad) == 0:
        return "shape/dtype mismatch"
    i = tuple(bad[0])
    return f"{len(bad)} mismatches; first at {i}: got {got[i]!r} want {want[i]!r}"


# ---- golden vectors recorded from the reference ---------------------------------------------
def test_golden_random(edt_gpu):
    for n, c in enumerate(load_golden("edt_random.npz")):
        lab = c["labels"]
        lab = np.asfortranarray(lab) if str(c["order"]) == "F" else np.ascontiguousarray(lab)
        an = tuple(c["anisotropy"])
        an = an[0] if lab.ndim == 1 else an
        bb = bool(c["black_border"])
        got = edt_gpu.edtsq(lab, anisotropy=an, black_border=bb)
        assert same(got, c["edtsq"]), (n, lab.shape, lab.dtype, explain(got, c["edtsq"]))
        got = edt_gpu.edt(lab, anisotropy=an, black_border=bb)
        assert same(got, c["edt"]), (n, lab.shape, lab.dtype, explain(got, c["edt"]))


def test_golden_configs(edt_gpu):
    for c in load_golden("edt_configs.npz"):
        lab = np.asfortranarray(c["labels"])
        got = edt_gpu.edtsq(lab, anisotropy=tuple(c["anisotropy"]), black_border=bool(c["black_border"]))
        assert same(got, c["edtsq"]), explain(got, c["edtsq"])


def test_golden_sdf_voxel_graph(edt_gpu):
    for c in load_golden("edt_sdf_voxel_graph.npz"):
        lab, an, bb = c["labels"], tuple(c["anisotropy"]), bool(c["black_border"])
        if str(c["kind"]) == "sdf":
            got = edt_gpu.sdf(lab, anisotropy=an, black_border=bb)
        else:
            got = edt_gpu.edtsq(lab, anisotropy=an, black_border=bb, voxel_graph=c["graph"])
        assert same(got, c["out"]), explain(got, c["out"])


# ---- randomized parity against the oracle -----------------------------------------------------
@pytest.mark.parametrize("seed", range(6))
def test_random_small_vs_oracle(edt_gpu, oracle_port, seed):
    rng = np.random.default_rng(100 + seed)
    for t in range(40):
        dims = int(rng.integers(1, 4))
        shape = tuple(int(rng.integers(1, 70)) for _ in range(dims))
        dtype = DTYPES[int(rng.integers(0, len(DTYPES)))]
        lab = blocky_labels(shape, nlabels=int(rng.integers(1, 9)), zero_frac=float(rng.random() * 0.5),
                            block=int(rng.integers(1, 9)), rng=rng).astype(dtype)
        if rng.random() < 0.5:
            lab = np.asfortranarray(lab)
        an = ANISO[int(rng.integers(0, len(ANISO)))][:dims]
        an = an[0] if dims == 1 else an
        bb = bool(rng.integers(0, 2))
        want = oracle_port.edtsq(lab, an, bb)
        got = edt_gpu.edtsq(lab, anisotropy=an, black_border=bb)
        assert same(got, want), (seed, t, shape, dtype, an, bb, explain(got, want))


@pytest.mark.parametrize("shape", [(1, 1, 1), (2, 1, 3), (1, 70, 1), (65, 33, 31), (64, 64, 64),
                                   (96, 80, 72), (130, 67, 35), (257, 5, 3), (3, 300, 2), (2, 3, 513)])
@pytest.mark.parametrize("bb", [False, True])
def test_odd_extents(edt_gpu, oracle_port, shape, bb):
    rng = np.random.default_rng(hash(shape) % 1000)
    lab = np.asfortranarray(blocky_labels(shape, 6, 0.15, 5, rng).astype(np.uint32))
    for an in ((1, 1, 1), (6, 6, 30)):
        want = oracle_port.edtsq(lab, an, bb)
        got = edt_gpu.edtsq(lab, anisotropy=an, black_border=bb)
        assert same(got, want), (shape, an, explain(got, want))
    want = oracle_port.edt(lab, (4, 4, 40), bb)
    got = edt_gpu.edt(lab, anisotropy=(4, 4, 40), black_border=bb)
    assert same(got, want), explain(got, want)  # 0 ULP


def test_long_rows_and_axes(edt_gpu, oracle_port):
    # rows / columns far longer than any LDS tile (cf. automated_test.py:819-823).  Rows of more than 2048 voxels go
    # through the line pipeline (one thread per voxel, csrc/edt_line.hip), not the thread-per-row fallback.
    rng = np.random.default_rng(5)
    for shape in ((5000, 3), (3, 5000), (46342, 1), (1, 46342)):
        lab = blocky_labels(shape, 4, 0.1, 37, rng).astype(np.float64)
        for bb in (False, True):
            want = oracle_port.edtsq(lab, (1.0, 2.0), bb)
            got = edt_gpu.edtsq(lab, anisotropy=(1.0, 2.0), black_border=bb)
            assert same(got, want), (shape, bb, explain(got, want))
            assert not np.any(np.isnan(got))
    # axes of 4097 .. 32735 rows: the workgroup-phased column kernel with 4 / 2 / 1 columns per tile (a thread per
    # band of every column); beyond that the size-agnostic kernels
    for shape in ((6, 9000), (3, 17000, 2), (2, 30000), (5, 4097, 3), (2, 33000)):
        lab = np.asfortranarray(blocky_labels(shape, 3, 0.1, int(rng.integers(20, 3000)), rng).astype(np.uint8))
        for an, bb in (((1.0, 1.0, 1.0)[:len(shape)], False), ((2.0, 0.5, 3.0)[:len(shape)], True)):
            want = oracle_port.edtsq(lab, an, bb)
            got = edt_gpu.edtsq(lab, anisotropy=an, black_border=bb)
            assert same(got, want), (shape, an, bb, explain(got, want))
    # 3-D volumes with long x rows: exact and tabulated (0.7: sequentially rounded sums) voxel sizes, runs that span
    # several 1024-voxel blocks, rows without any boundary (black_border off -> +inf / FLT_MAX before pass Y)
    for shape, dt in (((2500, 40, 6), np.uint32), ((4100, 9, 33), np.uint8), ((2049, 3, 2), np.uint16)):
        lab = np.asfortranarray(blocky_labels(shape, 3, 0.1, int(rng.integers(5, 900)), rng).astype(dt))
        lab[:, 0, 0] = 1   # one row, one label: no boundary along x at all
        for an in ((1.0, 1.0, 1.0), (0.7, 1.3, 2.0), (6.0, 6.0, 30.0)):
            for bb in (False, True):
                want = oracle_port.edtsq(lab, an, bb)
                got = edt_gpu.edtsq(lab, anisotropy=an, black_border=bb)
                assert same(got, want), (shape, an, bb, explain(got, want))


def test_extreme_anisotropy(edt_gpu, oracle_port):
    # automated_test.py:702-721, :791-817
    rng = np.random.default_rng(6)
    lab = blob_mask((40, 40, 40), rng=rng, p=0.7, block=5).astype(np.uint8)
    for an in ((1e6, 1.2e6, 40.0), (1e-7, 1e-7, 1e-7), (1e8, 1e8, 1e8), (1e-5, 1.0, 1e5), (0.001, 0.001, 0.001)):
        for bb in (False, True):
            want = oracle_port.edtsq(lab, an, bb)
            got = edt_gpu.edtsq(lab, anisotropy=an, black_border=bb)
            assert same(got, want), (an, bb, explain(got, want))


def test_float_label_semantics(edt_gpu, oracle_port):
    lab = np.array([[1.5, 1.5, -0.0, 2.0, np.nan, np.nan, 2.0, 2.0]] * 5, dtype=np.float32)
    for bb in (False, True):
        assert same(edt_gpu.edtsq(lab, black_border=bb), oracle_port.edtsq(lab, None, bb))
    lab64 = lab.astype(np.float64)
    assert same(edt_gpu.edtsq(lab64, black_border=True), oracle_port.edtsq(lab64, None, True))


def test_bool_equals_uint8_path(edt_gpu, oracle_port):
    rng = np.random.default_rng(7)
    m = blob_mask((33, 47, 29), rng=rng, p=0.5, block=4)
    for bb in (False, True):
        a = edt_gpu.edtsq(m.astype(bool), anisotropy=(2, 3, 5), black_border=bb)
        b = edt_gpu.edtsq(m.astype(np.uint8), anisotropy=(2, 3, 5), black_border=bb)
        assert same(a, b)
        assert same(a, oracle_port.edtsq(m.astype(bool), (2, 3, 5), bb))  # reference's binary route


def test_input_not_mutated_and_noncontiguous(edt_gpu, oracle_port):
    rng = np.random.default_rng(8)
    lab = blocky_labels((30, 40, 20), 5, 0.2, 4, rng).astype(np.uint16)
    keep = lab.copy()
    view = lab[::2, 1::3, :]
    got = edt_gpu.edtsq(view, anisotropy=(1, 2, 3))
    assert np.array_equal(lab, keep)
    assert same(got, oracle_port.edtsq(np.ascontiguousarray(view), (1, 2, 3), False))


def test_sdf_and_each(edt_gpu, oracle_port):
    rng = np.random.default_rng(9)
    lab = blocky_labels((24, 28, 20), 4, 0.4, 4, rng).astype(np.uint32)
    for bb in (False, True):
        assert same(edt_gpu.sdf(lab, anisotropy=(1, 1, 2), black_border=bb),
                    oracle_port.sdf(lab, (1, 1, 2), bb))
        assert same(edt_gpu.sdfsq(lab, anisotropy=(1, 1, 2), black_border=bb),
                    oracle_port.sdfsq(lab, (1, 1, 2), bb))
    dt = edt_gpu.edt(lab)
    seen = dict(edt_gpu.each(lab, dt))
    assert sorted(seen) == [k for k in np.unique(lab) if k]
    for k, img in seen.items():
        assert np.array_equal(img, dt * (lab == k))


def test_voxel_graph_3d(edt_gpu, oracle_port):
    rng = np.random.default_rng(10)
    m = blob_mask((20, 18, 22), rng=rng, p=0.8, block=3).astype(np.uint8)
    g = np.full(m.shape, 0b00111111, dtype=np.uint8)
    g[rng.random(m.shape) < 0.05] &= 0b11111110
    g[rng.random(m.shape) < 0.05] &= 0b11111011
    g[rng.random(m.shape) < 0.05] &= 0b11101111
    for bb in (False, True):
        for arr, gg in ((m, g), (np.asfortranarray(m), np.asfortranarray(g))):
            got = edt_gpu.edtsq(arr, anisotropy=(2, 2, 3), black_border=bb, voxel_graph=gg)
            want = oracle_port.edtsq(arr, (2, 2, 3), bb, voxel_graph=gg)
            assert same(got, want), explain(got, want)


# ---- BASELINE.json configurations at reduced size, bit-exact against the oracle ---------------
@pytest.mark.parametrize("cfg", ["cfg1", "cfg2", "cfg3", "cfg3m", "cfg5"])
def test_baseline_configs_reduced(edt_gpu, oracle_port, cfg):
    lab, an, bb = config_volume(cfg, 96)
    want = oracle_port.edtsq(lab, an, bb)
    got = edt_gpu.edtsq(lab, anisotropy=an, black_border=bb)
    assert same(got, want), explain(got, want)
    if cfg == "cfg5":
        assert same(edt_gpu.sdf(lab, anisotropy=an, black_border=bb), oracle_port.sdf(lab, an, bb))


# ---- full BASELINE sizes: size-independent properties ---------------------------------------
def test_cfg2_full_size_closed_form(edt_gpu):
    lab, an, bb = config_volume("cfg2", 512)
    got = edt_gpu.edtsq(lab, anisotropy=an, black_border=bb)
    assert got.max() == 2359296.0  # (6 * 256)^2, SURVEY 8(d)
    assert same(got, box_edtsq_closed_form(lab.shape, an))


def test_cfg3_full_size_vs_oracle_and_flip(edt_gpu, oracle_port):
    lab, an, bb = config_volume("cfg3", 512)
    got = edt_gpu.edtsq(lab, anisotropy=an, black_border=bb)
    assert len(np.unique(lab)) >= 1900
    # mirror symmetry: the transform commutes with flipping every axis
    flipped = np.asfortranarray(lab[::-1, ::-1, ::-1])
    got_f = edt_gpu.edtsq(flipped, anisotropy=an, black_border=bb)
    assert np.array_equal(got_f[::-1, ::-1, ::-1], got)
    want = oracle_port.edtsq(lab, an, bb)  # ~10 s single thread
    assert same(got, want), explain(got, want)


# ---- the two kernel families (LDS-tiled vs size-agnostic fallback) agree -----------------------
@pytest.mark.parametrize("seed", range(3))
def test_generic_path_matches_oracle(edt_gpu, oracle_port, seed, monkeypatch):
    monkeypatch.setenv("EDT_HIP_FORCE_GENERIC", "1")
    rng = np.random.default_rng(500 + seed)
    for t in range(30):
        dims = int(rng.integers(1, 4))
        shape = tuple(int(rng.integers(1, 80)) for _ in range(dims))
        lab = blocky_labels(shape, nlabels=int(rng.integers(1, 7)), zero_frac=float(rng.random() * 0.4),
                            block=int(rng.integers(1, 9)), rng=rng).astype(DTYPES[t % len(DTYPES)])
        an = ANISO[t % len(ANISO)][:dims]
        an = an[0] if dims == 1 else an
        bb = bool(t % 2)
        want = oracle_port.edtsq(lab, an, bb)
        got = edt_gpu.edtsq(lab, anisotropy=an, black_border=bb)
        assert same(got, want), (seed, t, shape, an, bb, explain(got, want))


def test_long_runs_deep_hulls(edt_gpu, oracle_port):
    """Wedges and cones: hulls hundreds of vertices deep and bridges far from the band seams."""
    n = 200
    x, y = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    shapes = {
        "wedge": (x + 2 * y < 3 * n // 2),
        "cone": ((x - n / 2) ** 2 + (y - n / 2) ** 2 < (n / 2.2) ** 2),
        "comb": ((x % 7 != 0) | (y > n // 2)),
        "stairs": ((x // 16) * 16 + 8 > y),
    }
    for name, m in shapes.items():
        vol = np.repeat(m[:, :, None], 40, axis=2).astype(np.uint8)
        if name in ("comb", "stairs"):
            vol[:, :, ::13] = 0  # cut the z columns into short runs
        for an in ((1, 1, 1), (3, 1, 2), (0.37, 1.9, 1.1)):
            for bb in (False, True):
                want = oracle_port.edtsq(vol, an, bb)
                got = edt_gpu.edtsq(vol, anisotropy=an, black_border=bb)
                assert same(got, want), (name, an, bb, explain(got, want))


def test_cpp_drop_in_header_on_gpu(edt_gpu, oracle_port, tmp_path):
    """The C++ facade (cpp/edt.hpp + cpp/edt_voxel_graph.hpp): fixed known answers, then randomized parity --
    cases written here from the CPU oracle, run by the C++ binary through the template of every label type
    (1-D / 2-D / 3-D, voxel graphs, caller-owned and library-allocated outputs) and compared bit for bit."""
    import struct
    import subprocess
    from test_abi import build_cpp_dropin
    from oracle import harness
    exe = build_cpp_dropin(tmp_path)
    res = subprocess.run([exe], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    rng = np.random.default_rng(2024)
    dtypes = [np.uint8, np.uint16, np.uint32, np.uint64, np.float32, np.float64, bool]
    blob = []
    for t in range(60):
        dims = 1 + t % 3
        dt = dtypes[t % len(dtypes)]
        shape = tuple(int(rng.integers(1, 40)) for _ in range(dims))
        lab = blocky_labels(shape, nlabels=1 if dt is bool else 5, zero_frac=0.25, block=int(rng.integers(1, 7)),
                            rng=rng).astype(dt)
        lab = np.asfortranarray(lab)                       # x fastest, as the C++ API takes it
        an = tuple(float(a) for a in ANISO[int(rng.integers(0, len(ANISO)))][:dims])
        bb = bool(rng.integers(0, 2))
        graph = None
        if dims >= 2 and t % 4 == 0:
            graph = np.asfortranarray(rng.integers(0, 64, size=shape).astype(np.uint8))
            graph[rng.random(shape) < 0.7] = 0b00111111
        want = oracle_port.edtsq(lab, an[0] if dims == 1 else an, bb, voxel_graph=graph)
        ext = shape + (1,) * (3 - dims)
        w = an + (1.0,) * (3 - dims)
        code = harness._DTYPE_CODE[np.dtype(dt)]
        blob.append(struct.pack("<7i3f", code, dims, ext[0], ext[1], ext[2], int(bb), int(graph is not None), *w))
        blob.append(lab.tobytes(order="F"))
        if graph is not None:
            blob.append(graph.tobytes(order="F"))
        blob.append(np.asfortranarray(want).astype(np.float32).tobytes(order="F"))
    # the binary route of the facade on MULTI-VALUED images (mode 2): labels split runs along x only
    # (src/edt.hpp:487-576, :681-755); expected values from the oracle's restatement of that route, which
    # tests/test_oracle.py pins to the compiled reference
    nbin = 0
    for t in range(28):
        dims = 2 + t % 2
        dt = dtypes[t % 6]                                 # every non-bool label type
        shape = tuple(int(rng.integers(1, 40)) for _ in range(dims))
        lab = np.asfortranarray(blocky_labels(shape, nlabels=4, zero_frac=0.3, block=int(rng.integers(1, 6)),
                                              rng=rng).astype(dt))
        an = tuple(float(a) for a in ANISO[int(rng.integers(0, len(ANISO)))][:dims])
        bb = bool(rng.integers(0, 2))
        want = oracle_port.binary_edtsq(lab, an, bb)
        ext = shape + (1,) * (3 - dims)
        w = an + (1.0,) * (3 - dims)
        blob.append(struct.pack("<7i3f", harness._DTYPE_CODE[np.dtype(dt)], dims, ext[0], ext[1], ext[2], int(bb), 2, *w))
        blob.append(lab.tobytes(order="F"))
        blob.append(np.asfortranarray(want).astype(np.float32).tobytes(order="F"))
        nbin += 1
    path = tmp_path / "cases.bin"
    path.write_bytes(struct.pack("<i", 60 + nbin) + b"".join(blob))
    res = subprocess.run([exe, str(path)], capture_output=True, text=True)
    assert res.returncode == 0 and f"{60 + nbin} of {60 + nbin}" in res.stdout, res.stdout + res.stderr


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.uint32, np.uint64, np.float32, np.float64])
def test_binary_route_multivalued_labels(edt_gpu, oracle_port, dtype):
    """edt_hip_binary_edtsq / EDT_FLAG_BINARY_YZ == pyedt::_binary_edt{2,3}dsq<T> (src/edt.hpp:487-576, :681-755) on
    images holding the values {0,1,2,3}: runs split by label along x only.  Against the compiled reference where it is
    built, else the oracle's restatement (pinned to it by tests/test_oracle.py).  Covers the wave kernels, the index
    form and the fp32 form of pass X, odd extents, both border modes and the fused sqrt."""
    from oracle import harness
    chk = harness.ref() if harness.have_ref() else oracle_port
    rng = np.random.default_rng(77)
    differs = 0
    for t, shape in enumerate([(37, 29, 31), (64, 64, 40), (96, 80), (33, 47), (130, 70, 36), (40, 300, 20)]):
        lab = blocky_labels(shape, nlabels=3, zero_frac=0.3, block=int(rng.integers(1, 6)), rng=rng).astype(dtype)
        if t % 2:
            lab = np.asfortranarray(lab)
        for an, bb in (((1, 1, 1), False), ((6, 6, 30), True), ((0.5, 0.7, 1.3), False)):
            an = an[:lab.ndim]
            want = chk.binary_edtsq(lab, an, bb)
            got = edt_gpu.binary_edtsq(lab, anisotropy=an, black_border=bb)
            assert same(got, want), (shape, an, bb, explain(got, want))
            assert same(edt_gpu.binary_edt(lab, anisotropy=an, black_border=bb), np.sqrt(want))
            differs += not same(want, edt_gpu.edtsq(lab, anisotropy=an, black_border=bb))
    assert differs >= 6  # the route really differs from the multi-label transform on these inputs
    # 0/1 input: both routes agree (and bool goes through the ordinary planes)
    img = (rng.random((50, 41, 33)) < 0.7).astype(dtype)
    assert same(edt_gpu.binary_edtsq(img, (6, 6, 30), True), edt_gpu.edtsq(img, (6, 6, 30), True))


def test_device_resident_voxel_graph(edt_gpu):
    """edt.device.edtsq_voxel_graph == the host entry point (which the golden vectors and the oracle pin)."""
    import torch
    from edt import device
    rng = np.random.default_rng(12)
    for shape, an, bb in (((37, 41, 29), (1.0, 2.0, 3.0), False), ((64, 48, 40), (6.0, 6.0, 30.0), True),
                          ((70, 55), (2.0, 1.0), True), ((33, 128), (1.0, 1.0), False)):
        lab = (rng.random(shape) < 0.7).astype(np.uint8) * rng.integers(1, 4, size=shape).astype(np.uint8)
        g = rng.integers(0, 64, size=shape).astype(np.uint8)
        g[rng.random(shape) < 0.6] = 0b00111111
        want = edt_gpu.edtsq(lab, anisotropy=an, black_border=bb, voxel_graph=g)
        got = device.edtsq_voxel_graph(torch.from_numpy(lab).cuda(), torch.from_numpy(g).cuda(), anisotropy=an,
                                       black_border=bb).cpu().numpy()
        assert np.array_equal(got, want, equal_nan=True), (shape, an, bb)
