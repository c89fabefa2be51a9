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
