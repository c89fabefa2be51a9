"""CPU-only checks of the drop-in boundary: the C-ABI library builds/loads, exports every
symbol include/edt_hip.h declares, validates arguments, and fails LOUDLY (no CPU fallback)
when there is no GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "edt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(edt_hip_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from edt import _lib
    return _lib.load()


def test_header_symbols_are_all_exported(lib):
    names = declared_symbols()
    assert len(names) >= 18
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/edt_hip.h but not exported"


def test_python_binding_covers_the_header():
    from edt import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_version_and_error_strings(lib):
    assert b"gfx950" in lib.edt_hip_version()
    assert isinstance(lib.edt_hip_last_error(), bytes)


def test_argument_validation_needs_no_gpu(lib):
    from edt import _lib
    # unknown dtype, bad ndim, unused extent != 1
    assert lib.edt_hip_workspace_bytes(99, 3, 4, 4, 4) == 0
    assert lib.edt_hip_workspace_bytes(_lib.U32, 4, 4, 4, 4) == 0
    assert lib.edt_hip_workspace_bytes(_lib.U32, 2, 4, 4, 4) == 0
    # the wave / tiled kernels work in place: scratch = four bit planes (1/8 byte per voxel each) + the 16-bit
    # distance indices of pass 1, one slab of at most 2^27 voxels of them (256 MiB) whatever the volume ...
    v = 64 * 64 * 64
    assert 4 * v // 8 + 2 * v <= lib.edt_hip_workspace_bytes(_lib.U32, 3, 64, 64, 64) <= 4 * v // 8 + 2 * v + 4096
    # (+ the hand-over list of the 16-bit integer column kernel: 4 bytes per tile)
    # ... unless every slab's indices fit the whole-volume limit (8 GiB of them; round 6: the 16-bit plane between passes Y and Z
    # then exists on volumes of several slabs too) -- 1024^3: 0.5 GiB of bit planes + 2 GiB of indices; beyond the limit: one slab
    # (+ 8 KiB per slice: slices of a whole multiple of 2 MiB lie that much further apart in the index buffer -- plane_pad_elems)
    assert lib.edt_hip_workspace_bytes(_lib.U32, 3, 1024, 1024, 1024) <= (1 << 29) + (1 << 31) + (1 << 19) + 1024 * 8192
    assert lib.edt_hip_workspace_bytes(_lib.U32, 3, 2048, 2048, 2048) <= (1 << 32) + (1 << 28) + (1 << 21)
    assert lib.edt_hip_workspace_bytes(_lib.U32, 3, 2048, 2048, 512) <= (1 << 32) + (1 << 30) + (1 << 21) + 512 * 8192   # (4 GiB of indices: within the limit)
    # ... which a caller can decline (EDT_FLAG_SMALL_WORKSPACE: fp32 between passes X and Y): 0.5 GiB for 1024^3
    assert lib.edt_hip_workspace_bytes_flags(_lib.U32, 3, 1024, 1024, 1024, _lib.FLAG_SMALL_WORKSPACE) <= (1 << 29) + (1 << 19)
    assert lib.edt_hip_workspace_bytes_flags(_lib.U32, 3, 64, 64, 64, _lib.FLAG_SMALL_WORKSPACE) <= 4 * v // 8 + 8192
    # (rows that are not whole 8-byte granules of indices keep the fp32 form of pass 1: bit planes only)
    assert lib.edt_hip_workspace_bytes(_lib.U32, 3, 63, 64, 64) <= 4 * 64 * 64 * 64 // 8 + 8192
    # ... only the size-agnostic kernels need a second fp32 volume and the hull stacks
    assert (lib.edt_hip_workspace_bytes_flags(_lib.U32, 3, 64, 64, 64, _lib.FLAG_FORCE_GENERIC)
            >= 2 * 64 * 64 * 64 * 4)
    buf = np.zeros(8, dtype=np.uint32)
    out = np.zeros(8, dtype=np.float32)
    rc = lib.edt_hip_squared_edt_1d_multi_seg(buf.ctypes.data, _lib.U32, out.ctypes.data, 8, 2, 1.0, 0)
    assert rc == -4  # EDT_ERR_UNSUPPORTED: stride != 1
    assert b"stride" in lib.edt_hip_last_error()


def test_empty_and_bad_shapes_python_level():
    import edt
    assert edt.edtsq(np.zeros((0,), dtype=np.uint8)).shape == (0,)
    assert edt.edt(np.zeros((4, 0, 3), dtype=np.uint32)).shape == (4, 0, 3)
    with pytest.raises(TypeError):
        edt.edtsq(np.zeros((2, 2, 2, 2), dtype=np.uint8))
    with pytest.raises(TypeError):
        edt.edtsq(np.zeros((5,), dtype=np.uint8), voxel_graph=np.zeros((5,), dtype=np.uint8))
    with pytest.raises(TypeError):
        edt.edtsq(np.zeros((5,), dtype=np.complex64))


def test_no_silent_cpu_fallback(lib):
    """Without a device every compute entry point must return EDT_ERR_NO_DEVICE."""
    from edt import _lib
    import edt
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present; the no-device path is exercised on the CPU runner")
    lab = np.ones((4, 4, 4), dtype=np.uint32)
    with pytest.raises(_lib.EdtHipError) as e:
        edt.edtsq(lab)
    assert e.value.code == _lib.ERR_NO_DEVICE
    with pytest.raises(_lib.EdtHipError):
        edt.sdf(lab)


def test_product_never_imports_the_oracle():
    """The shipped package must not import, link, load or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "euclidean-distance-transform-3d_amd")
    banned = re.compile(r"import\s+oracle|from\s+oracle|oracle/|libedt_oracle|libedt_ref|edt_oracle\.c|harness")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not banned.search(text), f"{os.path.join(dirpath, f)} references the oracle"


def build_cpp_dropin(tmp_path):
    import subprocess
    exe = str(tmp_path / "cpp_dropin")
    pkg = os.path.join(ROOT, "euclidean-distance-transform-3d_amd")
    cmd = ["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(pkg, "cpp"),
           os.path.join(ROOT, "tests", "cpp_dropin.cpp"), "-L" + os.path.join(pkg, "lib"), "-ledt_hip",
           "-Wl,-rpath," + os.path.join(pkg, "lib"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


def test_cpp_header_is_a_drop_in(tmp_path):
    """The reference's C++ signatures (edt::edt<T>, pyedt::_edt3dsq<T>, ...) compile and link
    against the C ABI; on a GPU-less host the call fails loudly instead of computing on the CPU."""
    import subprocess
    from edt import _lib
    exe = build_cpp_dropin(tmp_path)
    res = subprocess.run([exe], capture_output=True, text=True)
    if _lib.device_count() > 0:
        assert res.returncode == 0, res.stdout
    else:
        assert res.returncode == 3, res.stdout
        assert "no HIP device" in res.stdout


def test_index_form_criterion_of_the_library_implies_exact_sums(lib):
    """edt_hip_index_form_exact (the criterion the library uses to hand pass Y 16-bit distance indices): whenever it
    says yes, the reference's sequential fp32 sums of the voxel size (src/edt.hpp:97, :113) ARE the exact multiples;
    and it agrees with the mirror the lane-logic tests use."""
    from test_lane_logic import codes_exact
    rng = np.random.default_rng(11)
    yes = 0
    for t in range(6000):
        sx = int(rng.integers(1, 1100))
        kind = t % 5
        if kind == 0:
            w = np.float32(rng.integers(1, 40000) * 2.0 ** int(rng.integers(-40, 20)))
        elif kind == 1:
            w = np.float32(rng.integers(1, 64) / 8.0)
        elif kind == 2:
            w = np.float32(rng.uniform(0.01, 50.0))
        elif kind == 3:
            w = np.float32([0.0, -1.0, np.inf, np.nan, 1e-38, 3e38, 1e-31, 1e31][int(rng.integers(0, 8))])
        else:
            w = np.float32(2.0 ** 24 / (sx + 1) * rng.choice([0.5, 1.0, 1.0 + 2.0 ** -20]))
        got = lib.edt_hip_index_form_exact(float(w), sx)
        assert got == int(codes_exact(w, sx)), (w, sx)
        if not got:
            continue
        yes += 1
        if t % 3:
            continue  # (the sequential sums of every third accepted case)
        acc = np.float32(0)
        sums = np.empty(sx + 2, dtype=np.float32)
        for i in range(sx + 2):
            sums[i] = acc
            acc = np.float32(acc + w)
        assert np.array_equal(sums, np.arange(sx + 2, dtype=np.float32) * w), (w, sx)
    assert yes > 1500


def test_default_library_ignores_wrong_result_diagnostics():
    """The phase-skipping diagnostics (bits 1, 2, 4, 8, 0x200, 0x40000, 0x80000: wrong results, cost measurements) exist
    only in a -DEDT_DIAG build: the shipped library masks them out of whatever edt_hip_set_debug_mode /
    EDT_HIP_DEBUG_MODE asks for, keeps the form-selection bits, and the mode belongs to the calling thread."""
    import subprocess
    import sys
    import threading
    from edt import _lib
    lib = _lib.load()
    try:
        lib.edt_hip_set_debug_mode(0x200 | 0x80000 | 0x40000 | 15 | 0x4000 | 0x100000)
        assert lib.edt_hip_get_debug_mode() == 0x4000 | 0x100000
        seen = []
        t = threading.Thread(target=lambda: seen.append(lib.edt_hip_get_debug_mode()))
        t.start()
        t.join()
        assert seen == [0]          # another thread: its own mode
    finally:
        lib.edt_hip_set_debug_mode(0)
    code = ("import sys; sys.path.insert(0, %r); from edt import _lib; print(_lib.load().edt_hip_get_debug_mode())"
            % os.path.join(ROOT, "euclidean-distance-transform-3d_amd"))
    res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, EDT_HIP_DEBUG_MODE="0x8201"),
                         capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and res.stdout.strip() == str(0x8000), res.stdout + res.stderr


def test_field_floor_helper(lib):
    """edt_hip_field_floor(wx, wy) = min(fl32(wx*wx), fl32(wy*wy)): what the XY phase guarantees about every non-zero
    value it leaves (the `field_floor` argument of the sharded Z phase); 0 = unknown for sizes that are not positive and
    finite, and the shard entry points that take it refuse nothing for it (argument validation needs no GPU)."""
    f32 = np.float32
    for wx, wy in ((1.0, 1.0), (3.58, 40.0), (40.0, 3.58), (0.1, 0.3), (1e-3, 7.25)):
        want = min(float(f32(wx) * f32(wx)), float(f32(wy) * f32(wy)))
        assert lib.edt_hip_field_floor(wx, wy) == want
    for wx, wy in ((0.0, 1.0), (1.0, float("inf")), (float("nan"), 1.0), (1e30, 1.0), (1e-30, 1.0)):
        assert lib.edt_hip_field_floor(wx, wy) == 0.0
    # null pointers are still argument errors with the _ex forms
    assert lib.edt_hip_shard_z_device_ex(None, None, 8, 8, 8, 1.0, 1.0, 0, None, 0, None) == -2
    assert lib.edt_hip_shard_z_records_device_ex(None, 8, 8, 8, 1.0, 1.0, 0, None, 0, None) == -2


def test_records_of_16_bit_rows_host_side(lib):
    """edt_hip_shard_record16_words / edt_hip_shard_records16_supported are host arithmetic (no GPU): the record layout of the
    16-bit slab records (rows as packed pairs + the two bit planes, in 4-byte words) and where they apply -- voxel sizes that
    share a quantum, rows of whole 8-byte granules of indices, both scan axes of 97..1024 rows."""
    from edt import _lib
    words = lib.edt_hip_shard_record16_words
    assert words(1024, 128) == 128 * 1024 // 2 + 2 * 4 * 1024 == 73728          # 294 912 bytes against 557 056 of fp32 rows
    assert 4 * words(1024, 128) * 1.88 < 4 * lib.edt_hip_shard_record_floats(1024, 128)
    assert words(512, 40) == 40 * 512 // 2 + 2 * 2 * 512 and words(7, 32) == 0   # (odd rows have no packed form)
    ok = lib.edt_hip_shard_records16_supported
    assert ok(_lib.U32, 1024, 1024, 1024, 1.0, 1.0, 1.0) == 1 and ok(_lib.U8, 512, 512, 1024, 6.0, 6.0, 30.0) == 1
    assert ok(_lib.U32, 512, 512, 512, 3.58, 3.58, 40.0) == 0      # no quantum
    assert ok(_lib.U32, 512, 512, 64, 1.0, 1.0, 1.0) == 0          # z axis below four bands
    assert ok(_lib.U32, 512, 2048, 512, 1.0, 1.0, 1.0) == 0        # y axis beyond the integer kernel's 1024 rows
    assert ok(_lib.U32, 510, 512, 512, 1.0, 1.0, 1.0) == 0         # rows that are not whole granules
    assert ok(_lib.U32, 512, 512, 512, 0.7, 0.7, 0.7) == 0         # 0.7 k is not exact in fp32: no index form of pass X
    # the entry points refuse what they cannot serve before touching the device
    rc = lib.edt_hip_shard_z_records16_device(None, None, 512, 128, 512, 1.0, 1.0, 1.0, 0, None, 0, None)
    assert rc < 0 and b"null" in lib.edt_hip_last_error()


@pytest.mark.parametrize("bad", [-1.0, 0.0, float("nan"), float("inf"), -float("inf")])
def test_voxel_sizes_are_validated_before_any_device_work(lib, bad):
    """Zero / non-finite voxel sizes, and a negative one along x, are refused at the boundary (the reference does not
    validate them and its pass 1 crosses label boundaries for a negative size, src/edt.hpp:107-109 -- no kernel here answers
    that case, so every entry point says EDT_ERR_BAD_ARG instead of different kernels giving different fields).  A negative
    size along y or z is NOT refused: it enters the reference only as its square (src/edt.hpp:181, :258) and is
    taken as |w| -- such a call passes the validation and fails, here, only for want of a device.  The check comes before the
    device check: it can be exercised without a GPU."""
    import edt
    from edt import _lib
    lab = np.ones((4, 4, 4), dtype=np.uint32)
    out = np.empty(lab.size, dtype=np.float32)
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    sign_only = bad == -1.0
    for axis, w in enumerate(((bad, 1.0, 1.0), (1.0, bad, 1.0), (1.0, 1.0, bad))):
        rc = lib.edt_hip_edt3dsq(p(lab), _lib.U32, 4, 4, 4, w[0], w[1], w[2], 1, 1, p(out))
        if sign_only and axis > 0:
            assert rc != -2 or b"voxel size" not in lib.edt_hip_last_error()
        else:
            assert rc == -2 and b"voxel size" in lib.edt_hip_last_error()   # EDT_ERR_BAD_ARG
    # the Python layer: C order puts the fastest axis last
    for w in ((1.0, 1.0, bad),) + (() if sign_only else ((bad, 1.0, 1.0), (1.0, bad, 1.0))):
        with pytest.raises(ValueError):
            edt.edtsq(lab, anisotropy=w)
    with pytest.raises(ValueError):
        edt.edtsq(np.asfortranarray(lab), anisotropy=(bad, 1.0, 1.0))
    assert lib.edt_hip_squared_edt_1d_multi_seg(p(lab), _lib.U32, p(out), 64, 1, bad, 1) == -2
    rc = lib.edt_hip_sdf(p(lab), _lib.U32, 3, 4, 4, 4, 1.0, bad, 1.0, 1, 1, p(out))
    assert (rc != -2 or b"voxel size" not in lib.edt_hip_last_error()) if sign_only else rc == -2
    graph = np.zeros(lab.size, dtype=np.uint8)
    rc = lib.edt_hip_edt3dsq_voxel_graph(p(lab), _lib.U32, p(graph), 4, 4, 4, 1.0, 1.0, bad, 1, p(out))
    assert (rc != -2 or b"voxel size" not in lib.edt_hip_last_error()) if sign_only else rc == -2
    # (an unused axis is not looked at)
    with pytest.raises(ValueError):
        edt.edt(np.ones((4, 4), dtype=np.uint8), anisotropy=(1.0, bad))


def test_import_probe_says_why_on_a_host_without_a_gpu():
    """VERDICT r5 "What's missing" 4: a host that cannot run any transform fails at IMPORT, as ImportError carrying the reason,
    not at the first call -- unless EDT_HIP_ALLOW_NO_DEVICE=1 (this tier sets it: tests/conftest.py).  The probe does not
    initialise the HIP runtime (fork-safe): it looks for the built library and the KFD device node."""
    import subprocess
    import sys
    if os.path.exists("/dev/kfd"):
        pytest.skip("this host has a GPU device node: the import succeeds")
    pkg = os.path.join(ROOT, "euclidean-distance-transform-3d_amd")
    env = {k: v for k, v in os.environ.items() if k != "EDT_HIP_ALLOW_NO_DEVICE"}
    env["PYTHONPATH"] = pkg
    res = subprocess.run([sys.executable, "-c", "import edt"], env=env, capture_output=True, text=True, timeout=120, cwd="/tmp")
    assert res.returncode != 0 and "ImportError" in res.stderr and "/dev/kfd" in res.stderr and "no CPU fallback" in res.stderr
    env["EDT_HIP_ALLOW_NO_DEVICE"] = "1"
    res = subprocess.run([sys.executable, "-c", "import edt; print(edt.edtsq.__name__)"], env=env, capture_output=True, text=True,
                         timeout=120, cwd="/tmp")
    assert res.returncode == 0 and "edtsq" in res.stdout, res.stderr[-1500:]


def test_pitch_of_the_index_buffer_follows_the_slice_size(lib, monkeypatch):
    """Round 6 (csrc/edt_api.hip: plane_pad_elems): the slices of the index buffer / 16-bit plane lie 8 KiB further apart where a slice
    (2 bytes per voxel) is a whole multiple of 2 MiB, 4 KiB where of 1 MiB, and sx * sy elements apart everywhere else -- seen from
    outside as workspace bytes (host arithmetic: no GPU).  EDT_HIP_PLANE_PAD_BYTES overrides per plan; a 2-D call and a stack of
    images (EDT_FLAG_BATCH_2D) have no plane between passes Y and Z and no pad."""
    from edt import _lib
    monkeypatch.delenv("EDT_HIP_PLANE_PAD_BYTES", raising=False)

    def ws(shape, flags=0, ndim=3):
        return lib.edt_hip_workspace_bytes_flags(_lib.U32, ndim, *shape, flags)

    def pad_per_slice(shape):
        monkeypatch.setenv("EDT_HIP_PLANE_PAD_BYTES", "0")
        without = ws(shape)
        monkeypatch.delenv("EDT_HIP_PLANE_PAD_BYTES")
        extra = ws(shape) - without
        assert extra % shape[2] == 0 or extra < 4096, (shape, extra)   # (the carver rounds every buffer up to its alignment)
        return round(extra / shape[2])

    assert pad_per_slice((1024, 1024, 64)) == 8192      # 2 MiB slices
    assert pad_per_slice((2048, 512, 40)) == 8192
    assert pad_per_slice((2048, 2048, 8)) == 8192       # 8 MiB
    assert pad_per_slice((1024, 512, 64)) == 4096       # 1 MiB
    assert pad_per_slice((512, 1024, 64)) == 4096
    assert pad_per_slice((512, 512, 512)) == 0          # 512 KiB: best as it is
    assert pad_per_slice((1024, 1008, 64)) == 0
    assert pad_per_slice((640, 512, 64)) == 0
    monkeypatch.setenv("EDT_HIP_PLANE_PAD_BYTES", "4096")
    forced = ws((160, 300, 140))
    monkeypatch.setenv("EDT_HIP_PLANE_PAD_BYTES", "0")
    assert 140 * 4096 <= forced - ws((160, 300, 140)) < 140 * 4096 + 4096
    # no plane, no pad: two dimensions, a stack of images
    monkeypatch.setenv("EDT_HIP_PLANE_PAD_BYTES", "8192")
    with_pad_2d = ws((1024, 1024, 1), ndim=2)
    with_pad_stack = ws((1024, 1024, 16), flags=_lib.FLAG_BATCH_2D)
    monkeypatch.setenv("EDT_HIP_PLANE_PAD_BYTES", "0")
    assert with_pad_2d == ws((1024, 1024, 1), ndim=2)
    assert with_pad_stack == ws((1024, 1024, 16), flags=_lib.FLAG_BATCH_2D)
