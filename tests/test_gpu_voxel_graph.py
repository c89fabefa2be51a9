"""GPU tier: the native voxel-graph transform (csrc/edt_voxel_graph.hip: pruned doubled grid, no 8x volume)
against the CPU oracle's restatement of the reference's up-sampled formulation
(src/edt_voxel_graph.hpp:54-117, :120-214), against this library's own up-sampled fallback (debug bit
0x20000), and -- cfg5 at 512^3, the whole volume -- against the compiled reference's own voxel-graph transform."""
import os

import numpy as np
import pytest

from synth import blob_mask, config_volume

pytestmark = pytest.mark.gpu


def explain_vg(got, want):
    bad = np.argwhere(got != want)
    return f"{len(bad)} mismatches; first at {tuple(bad[0])}: got {got[tuple(bad[0])]!r} want {want[tuple(bad[0])]!r}" if len(bad) else "equal"


def _graph(shape, rng, p):
    g = np.full(shape, 0b00111111, dtype=np.uint8)
    for bit in (0x01, 0x04, 0x10):
        g[rng.random(shape) < p] &= np.uint8(~bit & 0xFF)
    return g


@pytest.mark.parametrize("seed", range(8))
def test_native_vs_oracle_and_upsampled_fallback(edt_gpu, oracle_port, seed):
    from edt import _lib
    lib = _lib.load()
    rng = np.random.default_rng(500 + seed)
    for t in range(10):
        dims = 2 if t % 3 == 0 else 3
        shape = tuple(int(rng.integers(1, 44)) for _ in range(dims))
        if t == 9:
            shape = (70, 3, 33)[:dims]
        dtype = [np.uint8, np.uint16, np.uint32, np.uint64, np.float32, bool][int(rng.integers(0, 6))]
        m = blob_mask(shape, rng=rng, p=float(rng.uniform(0.5, 1.0)), block=int(rng.integers(1, 6)))
        lab = (m * rng.integers(1, 5, size=shape)).astype(dtype)
        g = _graph(shape, rng, float(rng.uniform(0.0, 0.15)))
        if rng.random() < 0.5:
            lab, g = np.asfortranarray(lab), np.asfortranarray(g)
        an = [(1.0, 1.0, 1.0), (2.0, 2.0, 3.0), (6.0, 6.0, 30.0), (0.7, 1.3, 2.1)][int(rng.integers(0, 4))][:dims]
        bb = bool(rng.integers(0, 2))
        want = oracle_port.edtsq(lab, an, bb, voxel_graph=g)
        got = edt_gpu.edtsq(lab, anisotropy=an, black_border=bb, voxel_graph=g)
        assert got.shape == want.shape and np.array_equal(got, want, equal_nan=True), (seed, t, shape, dtype, an, bb)
        lib.edt_hip_set_debug_mode(0x20000)  # the up-sampled formulation of this library
        try:
            old = edt_gpu.edtsq(lab, anisotropy=an, black_border=bb, voxel_graph=g)
        finally:
            lib.edt_hip_set_debug_mode(0)
        assert np.array_equal(old, want, equal_nan=True), (seed, t, "fallback")
        assert np.array_equal(edt_gpu.edt(lab, anisotropy=an, black_border=bb, voxel_graph=g), np.sqrt(want),
                              equal_nan=True)


@pytest.mark.parametrize("shape", [(512, 40, 6), (9, 600, 5), (12, 7, 520), (260, 260, 3)])
def test_native_long_axes(edt_gpu, oracle_port, shape):
    """doubled axes of 1024+ rows: the 2- and 1-column wave shapes of the column kernel"""
    rng = np.random.default_rng(sum(shape))
    lab = blob_mask(shape, rng=rng, p=0.85, block=9).astype(np.uint8)
    g = _graph(shape, rng, 0.02)
    for bb in (True, False):
        want = oracle_port.edtsq(lab, (1.0, 2.0, 1.5), bb, voxel_graph=g)
        got = edt_gpu.edtsq(lab, anisotropy=(1.0, 2.0, 1.5), black_border=bb, voxel_graph=g)
        assert np.array_equal(got, want, equal_nan=True), (shape, bb)


def test_cfg5_512_device_resident_workspace_and_parity(edt_gpu, oracle_ref):
    """BASELINE configs[4] at FULL size against the compiled reference's own voxel-graph transform
    (pyedt::_edt3dsq_voxel_graph, src/edt_voxel_graph.hpp:120-214 -- single-threaded upstream, about a minute at
    512^3): scratch without any 8x temporary, the native device-resident form bit-identical to the reference, and
    the up-sampled fallback formulation of this library identical to both."""
    import torch
    from edt import _lib, device
    lib = _lib.load()
    lab, an, bb = config_volume("cfg5", 512)
    rng = np.random.default_rng(55)
    g = _graph(lab.shape, rng, 0.01)
    vox = lab.size
    need = lib.edt_hip_voxel_graph_workspace_bytes(3, *lab.shape)
    assert need < 4.4 * vox * 4, need        # 4 x voxels floats + bit planes, not 8 x voxels x (1 + 4) bytes
    tl = torch.from_numpy(np.ascontiguousarray(lab.T)).cuda()
    tg = torch.from_numpy(np.ascontiguousarray(g.T)).cuda()
    got = device.edtsq_voxel_graph(tl, tg, anisotropy=an[::-1], black_border=bb)
    torch.cuda.synchronize()
    want = oracle_ref.edtsq(lab, an, bb, voxel_graph=g)
    got_np = got.cpu().numpy().T
    assert np.array_equal(got_np, want), explain_vg(got_np, want)
    lib.edt_hip_set_debug_mode(0x20000)  # (form selection, this thread only: the up-sampled formulation)
    try:
        old = device.edtsq_voxel_graph(tl, tg, anisotropy=an[::-1], black_border=bb)
        torch.cuda.synchronize()
    finally:
        lib.edt_hip_set_debug_mode(0)
    assert torch.equal(got, old)
