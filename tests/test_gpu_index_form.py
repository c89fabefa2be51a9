"""GPU tier: the index form of pass 1 (pass X hands pass Y 16-bit distance indices where every multiple of the voxel
size is exact in fp32; csrc/edt_rowwave.hip C16 -> csrc/edt_colwave_kernel.h XF) against the oracle -- at the edges
of its exactness criterion, for rows it does not serve, for squares that overflow, for rows without any boundary --
and against the fp32 form of the same library (debug bit 0x100000)."""
import numpy as np
import pytest

from synth import blocky_labels

pytestmark = pytest.mark.gpu

# voxel sizes along x: exact multiples up to the row length (index form), just not (fp32 form), tiny, squares that overflow
WX = [16381.0, 16383.0, 2.0 ** -60, 3.0 * 2.0 ** 40, 1.0e19, 0.7, 0.375]
SHAPES = [(1024, 40, 3), (1000, 33, 2), (516, 70, 5), (130, 64, 9), (64, 300, 4), (8, 8, 1100), (512, 96),
          (8, 12, 4200), (16, 2100, 3)]  # (z beyond the in-place kernels: index form + ping-pong z pass; y beyond the wave kernel)


def _labels(shape, kind, rng):
    if kind == "ones":
        return np.ones(shape, dtype=np.uint16)
    if kind == "rows":  # whole rows of one label: without a black border neither side has a boundary
        lab = np.ones(shape, dtype=np.uint8)
        lab[:, ::3] = 2
        lab[:, 1::7] = 0
        return lab
    return blocky_labels(shape, nlabels=5, zero_frac=0.1, block=int(rng.integers(2, 30)), rng=rng).astype(np.uint32)


@pytest.mark.parametrize("shape", SHAPES)
def test_index_form_edges_match_the_oracle_and_the_fp32_form(edt_gpu, oracle_port, shape):
    from edt import _lib
    lib = _lib.load()
    rng = np.random.default_rng(sum(shape))
    for kind in ("ones", "rows", "blocky"):
        lab = np.asfortranarray(_labels(shape, kind, rng))
        for wx in WX:
            an = (wx, 2.0, 0.5)[:len(shape)]
            for bb in (True, False):
                want = oracle_port.edtsq(lab, an, bb)
                got = edt_gpu.edtsq(lab, anisotropy=an, black_border=bb)
                assert np.array_equal(got, want), (shape, kind, wx, bb)
                lib.edt_hip_set_debug_mode(0x100000)
                try:
                    plain = edt_gpu.edtsq(lab, anisotropy=an, black_border=bb)
                finally:
                    lib.edt_hip_set_debug_mode(0)
                assert np.array_equal(plain, want), (shape, kind, wx, bb, "fp32 form")


def test_workspace_sized_without_the_index_buffer_still_serves(edt_gpu, oracle_port):
    """a plan made while the index form was switched off has no room for the indices: the call takes the fp32 form"""
    import torch
    from edt import _lib, device
    lib = _lib.load()
    rng = np.random.default_rng(9)
    shape = (256, 96, 40)
    lab = np.asfortranarray(blocky_labels(shape, nlabels=6, zero_frac=0.1, block=9, rng=rng).astype(np.uint32))
    want = oracle_port.edtsq(lab, (6.0, 6.0, 30.0), False)
    t = torch.from_numpy(np.ascontiguousarray(lab.T).view(np.int32)).cuda()
    full = lib.edt_hip_workspace_bytes(_lib.U32, 3, *shape)
    lib.edt_hip_set_debug_mode(0x100000)
    try:
        small_plan = device.Plan(shape, _lib.U32)
        assert small_plan.workspace.numel() < full - 2 * lab.size + 4096
    finally:
        lib.edt_hip_set_debug_mode(0)
    got = small_plan.run(t, (6.0, 6.0, 30.0), black_border=False).cpu().numpy().T
    assert np.array_equal(got, want)
    lean_plan = device.Plan(shape, _lib.U32, small_workspace=True)  # the public way to decline the index buffer
    assert lean_plan.workspace.numel() == small_plan.workspace.numel()
    got = lean_plan.run(t, (6.0, 6.0, 30.0), black_border=False).cpu().numpy().T
    assert np.array_equal(got, want)
    big_plan = device.Plan(shape, _lib.U32)
    assert big_plan.workspace.numel() == full
    got = big_plan.run(t, (6.0, 6.0, 30.0), black_border=False).cpu().numpy().T
    assert np.array_equal(got, want)


def test_slab_by_slab_passes_on_a_volume_larger_than_the_index_buffer(edt_gpu, oracle_ref):
    """more than 2^27 voxels: passes X and Y run slab by slab over one buffer of indices (halo slice between slabs)"""
    import os
    from synth import voronoi_coarse
    coarse = voronoi_coarse((128, 128, 160), nseeds=900, seed=3)
    lab = np.asfortranarray(coarse.repeat(4, 0).repeat(4, 1).repeat(4, 2))  # 512 x 512 x 640: slabs of 512 + 128 slices
    lab[:, :, 510:513] = 0  # a zero sheet across the slab boundary is part of the z structure
    want = oracle_ref.edtsq(lab, (6.0, 6.0, 30.0), False, parallel=os.cpu_count())
    got = edt_gpu.edtsq(lab, anisotropy=(6.0, 6.0, 30.0), black_border=False)
    assert np.array_equal(got, want)
