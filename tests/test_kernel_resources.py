"""CPU tier (needs hipcc only): the column-pass kernels are built WITHOUT scratch traffic in their bodies and at
four waves per SIMD.

The hull path sits at the edge of the 128-VGPR budget; a change anywhere in the kernel body once tipped its hot
loops into scratch with byte-identical hull code (cfg2 0.69 -> 1.05 ms).  tools/check_spills.py counts the
scratch_load / scratch_store instructions of the compiled kernels; this test pins the two shapes the BASELINE
volumes use (512-row axes: 4-column waves; 1024-row axes: 2-column waves)."""
import os
import shutil
import sys
import tempfile

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def scans():
    """both shapes compiled side by side (device code only)"""
    if not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    import check_spills
    from concurrent.futures import ThreadPoolExecutor
    with tempfile.TemporaryDirectory() as d, ThreadPoolExecutor(2) as ex:
        res = list(ex.map(lambda cw: check_spills.scan(cw, d), (4, 2)))
    return dict(zip((4, 2), res))


@pytest.mark.parametrize("cw", [4, 2])
def test_column_kernels_have_no_scratch_in_their_bodies(scans, cw):
    funcs = scans[cw]
    kernels = [f for f in funcs if f["kernel"]]
    assert len(kernels) == 8  # border rule x {fp32, index form of pass 1} x {in place, slab records}
    for f in kernels:
        assert f["scratch_ops"] <= 8, f
        assert f["occupancy"] is None or f["occupancy"] >= 4, f
        assert f["vgprs"] is None or f["vgprs"] <= 128, f
    # (no non-inlined device functions in the default build: a call would push callee-saved registers through scratch)
    assert all(f["kernel"] for f in funcs), [f["name"] for f in funcs if not f["kernel"]]


def test_integer_column_kernels_have_no_scratch_and_four_waves_per_simd():
    """csrc/edt_colq16.hip with the wide form inlined (round 5): every instantiation free of scratch instructions, at most
    128 VGPRs (four workgroups of four waves per CU is what the 39 KiB image was sized for)."""
    if not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    import q16_resources
    funcs = q16_resources.scan()
    assert len(funcs) >= 30
    for f in funcs:
        assert f["scratch_ops"] == 0, f
        assert f["vgprs"] <= 128 and (f["occupancy"] is None or f["occupancy"] >= 4), f
    # Round 6: the fill of a tile is ONE trip to memory -- the sixteen 8-byte row loads of a thread (index form: IN = 1; the 16-bit
    # plane: IN = 2) leave back to back, no `s_waitcnt vmcnt` between them.  The source cannot show this: the build before had the
    # same sixteen loads in its text and a wait behind the second one in its ISA (a register copy of the allocator's), and in pass Z
    # a scalar load and its wait ahead of each -- profiles/r06_fill_ab.txt.  Pinned here so that a compiler or a well-meant edit
    # that brings a branch back between the loads shows up without a GPU.
    import re
    seen = 0
    for f in funcs:
        m = re.search(r"k_column_pass_q16<(?:true|false), (\d),", f["name"])
        if m and m.group(1) in ("1", "2"):
            seen += 1
            assert f["loads_in_flight"] >= 16, f
    assert seen >= 20


def test_traffic_table_names_kernels_of_this_build():
    """bench.py prices the kernels' real traffic with a table from a committed profile (profiles/<tag>_traffic.json: PMC
    passes of a profiling run, not re-measured in the bench run).  Nothing used to fail when the kernels changed and the
    table did not: every kernel the table names must be a kernel of the library as built now --
    a renamed / re-templated / removed kernel means the profile has to be taken again."""
    import json
    import re
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    lib = os.path.join(ROOT, "euclidean-distance-transform-3d_amd", "lib", "libedt_hip.so")
    if not os.path.exists(lib) or shutil.which("c++filt") is None or shutil.which("strings") is None:
        pytest.skip("library not built / binutils missing")
    table = os.path.join(ROOT, "profiles", f"{bench.PROFILE_TAG}_traffic.json")
    if not os.path.exists(table):
        pytest.skip(f"{table}: this round's profile has not been taken yet (bench.py then reports no real_* fields)")
    mangled = sorted(set(re.findall(r"_ZN7edt_amd\w+", subprocess.run(["strings", "-n", "16", lib], capture_output=True, text=True).stdout)))
    dem = subprocess.run(["c++filt"], input="\n".join(mangled), capture_output=True, text=True).stdout.splitlines()
    have = {re.sub(r"\(.*", "", d.replace("(anonymous namespace)::", "")).replace("void ", "").replace("edt_amd::", "") for d in dem}
    blob = json.load(open(table))
    named = set()
    for cfg, passes in blob.items():
        if not isinstance(passes, dict):
            continue
        for p, entry in passes.items():
            if isinstance(entry, dict):
                named.update(k for k in entry.get("kernels", []) if not k.startswith("__amd_rocclr"))
    assert named, "the table names no kernel"
    missing = sorted(k for k in named if k not in have)
    assert not missing, f"{table} names kernels this build does not have (profile again): {missing}"


def test_traffic_table_names_no_launch_the_build_does_not_make():
    """round 5's committed kernel statistics of cfg3 / cfg3L still listed the list-mode launches of the
    fp32 kernel and the memset of the hand-over counters, which that round's final build no longer made there.  Where the
    library's own host proof (edt_hip_q16_no_refusals: host arithmetic, no device) says a configuration's column passes can
    refuse no tile, the build launches neither -- so the committed table of this round must not name them for that
    configuration: the fp32 column kernel (`k_column_pass_wave`) under y_pass / z_pass, or the runtime's fill kernel
    ("other": the memset)."""
    import ctypes
    import json
    sys.path.insert(0, ROOT)
    import bench
    from edt import _lib
    from synth import SWEEP
    table = os.path.join(ROOT, "profiles", f"{bench.PROFILE_TAG}_traffic.json")
    lib_path = os.path.join(ROOT, "euclidean-distance-transform-3d_amd", "lib", "libedt_hip.so")
    if not os.path.exists(table) or not os.path.exists(lib_path):
        pytest.skip("this round's profile has not been taken yet / library not built")
    lib = _lib.load()
    # what the profiled configurations are: extents, voxel sizes, border rule (tests/synth.py: config_volume, SWEEP)
    cfgs = {"cfg1": ((512,) * 3, (1.0, 1.0, 1.0), True), "cfg2": ((512,) * 3, (6.0, 6.0, 30.0), True),
            "cfg3": ((512,) * 3, (1.0, 1.0, 1.0), False), "cfg3m": ((512,) * 3, (6.0, 6.0, 30.0), False),
            "cfg3L": ((512,) * 3, (1.0, 1.0, 1.0), False), "cfg3M": ((512,) * 3, (1.0, 1.0, 1.0), False),
            "cfg3La": ((512,) * 3, (6.0, 6.0, 30.0), False), "cfg3Ma": ((512,) * 3, (6.0, 6.0, 30.0), False),
            "cfg4": ((1024,) * 3, (1.0, 1.0, 1.0), False)}
    for name, (kind, par, an, bb) in SWEEP.items():
        cfgs[name] = ((1024, 1024, 128) if kind == "sphere_slab" else (512,) * 3, an, bb)
    blob = json.load(open(table))
    checked = 0
    for cfg, passes in blob.items():
        if cfg not in cfgs or not isinstance(passes, dict):
            continue
        (sx, sy, sz), w, bb = cfgs[cfg]
        y, z = ctypes.c_int(-1), ctypes.c_int(-1)
        rc = lib.edt_hip_q16_no_refusals(sx, sy, sz, w[0], w[1], w[2], 3, int(bb), ctypes.addressof(y), ctypes.addressof(z))
        if rc != 1:
            continue
        for p, sure in (("y_pass", bool(y.value)), ("z_pass", bool(z.value))):
            kernels = passes.get(p, {}).get("kernels", [])
            if sure:
                assert not [k for k in kernels if "k_column_pass_wave" in k], (cfg, p, kernels)
            checked += 1
        if y.value and z.value:
            assert "other" not in passes or passes["other"]["total"] == 0, (cfg, "a memset / fill the build does not launch", passes["other"])
    assert checked > 0
