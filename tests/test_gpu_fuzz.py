"""Seeded slices of the randomised parity runs (tools/fuzz_gpu.py, fuzz_shard.py, fuzz_driver.py) inside the `-m gpu` tier
(VERDICT r5 "What's weak" 1: the widest fuzz lived in tools/ and the driver's tier never exercised it).  Every slice is a
subprocess with a fixed seed -- the tools are what a builder runs with thousands of cases; here ~90 s in all:

  * general shapes / dtypes / orders / label structures;
  * the integer column kernel's shapes and voxel sizes (FUZZ_Q16=1) under the default form selection, with tiles beyond 16
    bits always as two wide passes (0x40000000), without the wide form (0x20000000: the fp32 hand-over), with fp32 values
    between passes Y and Z (0x10000000), and on volumes of +inf (FUZZ_INF=1: one object with sparse structure and no border --
    tiles answered from the fill, +inf rows in the 16-bit plane, windows that start and stop at the finite rows, refused tiles
    with such rows: the class of the one mismatch round 6's closing fuzz found), on volumes of slabs and boxes (FUZZ_FLAT=1: tiles
    without structure along the scan axis, answered from their image) and with a random pitch of the index buffer / 16-bit
    plane per case (FUZZ_PAD=1: EDT_HIP_PLANE_PAD_BYTES unset / 0 / 8 / 4096 / 8200);
  * the voxel-graph transform (FUZZ_VG=1);
  * the two sharded phases as virtual ranks (16-bit and fp32 slab records);
  * the whole sharded driver as two processes sharing the GPU over gloo.
GPU vs the CPU oracle, bit for bit; a mismatch fails the slice and prints the tool's MISMATCH lines."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, args, env=None, timeout=900):
    e = dict(os.environ)
    e.pop("EDT_HIP_DEBUG_MODE", None)
    e.update(env or {})
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), *[str(a) for a in args]], cwd=ROOT, env=e,
                         capture_output=True, text=True, timeout=timeout)
    tail = (res.stdout + res.stderr)[-3000:]
    assert res.returncode == 0, tail
    assert "MISMATCH" not in res.stdout, tail
    return res.stdout


@pytest.mark.parametrize("ncases,seed,env", [
    (150, 601, {}),
    (80, 602, {"FUZZ_Q16": "1"}),
    (40, 603, {"FUZZ_Q16": "1", "EDT_HIP_DEBUG_MODE": "0x40000000"}),
    (30, 604, {"FUZZ_Q16": "1", "EDT_HIP_DEBUG_MODE": "0x20000000"}),
    (30, 605, {"FUZZ_Q16": "1", "EDT_HIP_DEBUG_MODE": "0x10000000"}),
    (60, 608, {"FUZZ_Q16": "1", "FUZZ_INF": "1"}),
    (60, 609, {"FUZZ_Q16": "1", "FUZZ_FLAT": "1", "FUZZ_PAD": "1"}),
    (40, 610, {"FUZZ_Q16": "1", "FUZZ_PAD": "1"}),
    (80, 606, {"FUZZ_VG": "1"}),
], ids=["general", "q16", "q16_two_wide_passes", "q16_no_wide_form", "q16_fp32_between_y_and_z", "q16_volumes_of_inf",
        "q16_slabs_and_boxes_padded_pitch", "q16_padded_pitch", "voxel_graph"])
def test_fuzz_gpu_slice(edt_gpu, oracle_port, ncases, seed, env):
    out = _run("fuzz_gpu.py", [ncases, seed], env)
    assert f"{ncases} cases, 0 mismatches" in out, out[-500:]


def test_fuzz_sharded_phases_slice(edt_gpu, oracle_port):
    out = _run("fuzz_shard.py", [100, 611])
    assert "100 cases, 0 mismatches" in out, out[-500:]


def test_fuzz_sharded_driver_two_processes_slice(edt_gpu, oracle_port):
    out = _run("fuzz_driver.py", [2, 60, 612])
    assert "0 mismatching" in out, out[-500:]
