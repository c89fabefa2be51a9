"""Full-size parity on the GPU (`-m gpu`): the BASELINE volumes against the COMPILED REFERENCE itself
(oracle/_ref, built from the reference's sources by oracle/Makefile; it travels to the GPU box), run with
every host thread -- not against our restatement.

  * cfg3 / cfg3m (configs[2]): 512^3 uint32, ~2000 labels, black_border=False, (1,1,1) and (6,6,30)
  * cfg4 (configs[3]) on ONE GPU: the 1024^3 segmentation
  * the sdf leg of configs[4]: sdf / sdfsq of the 512^3 uint8 blob volume (device-resident and through the host entry point)
  * cfg4-sized virtual ranks: 1024 x 1000 x 1016 cut into 8 uneven Z-slabs / Y-slabs,
    every rank's two phases on one device, the "exchange" a device copy
  * two host threads driving two streams with their own plans at the same time
"""
import os
import threading

import numpy as np
import pytest

from synth import blocky_labels, config_volume, several_slabs_volume, voronoi_labels

pytestmark = pytest.mark.gpu


def _ref_edtsq(oracle_ref, lab, an, bb):
    return oracle_ref.edtsq(lab, an, bb, parallel=os.cpu_count() or 1)


@pytest.mark.parametrize("name", ["cfg3", "cfg3m"])
def test_cfg3_512_against_compiled_reference(edt_gpu, oracle_ref, name):
    import torch
    from edt import device

    lab, an, bb = config_volume(name, 512)
    want = _ref_edtsq(oracle_ref, lab, an, bb)
    t = torch.from_numpy(np.ascontiguousarray(lab.T).view(np.int32)).cuda()
    got = device.edtsq(t, anisotropy=an[::-1], black_border=bb).cpu().numpy().T
    assert np.array_equal(got, want)
    # the host-buffer entry point (C ABI, numpy in / numpy out) on the same volume
    assert np.array_equal(edt_gpu.edtsq(lab, anisotropy=an, black_border=bb), want)


@pytest.mark.parametrize("name", ["sw256", "sphere250", "onesF", "onebg", "diagF", "sphere_slab"])
def test_object_sweep_against_compiled_reference(edt_gpu, oracle_ref, name):
    """The object-size sweep of bench.py (tests/synth.py: SWEEP) as parity cases at full size: LARGE objects -- cells ~256 voxels
    across, one ball of radius 250, a box without any boundary (+inf everywhere), a box with ONE background voxel (every
    z-column one finite row: the windows of +inf rows, round 6), half spaces cut diagonally without a border (values beyond 16
    bits AND rows without a boundary in every tile: the wide form throughout), the ball on the 8-GPU slab shape -- against the
    compiled reference with every host thread, bit for bit; with and without the fused sqrt."""
    import torch
    from edt import device

    lab, an, bb = config_volume(name, 512)
    want = _ref_edtsq(oracle_ref, lab, an, bb)
    t = torch.from_numpy(np.ascontiguousarray(lab.T).view(np.int32)).cuda()
    got = device.edtsq(t, anisotropy=an[::-1], black_border=bb).cpu().numpy().T
    assert np.array_equal(got, want)
    got = device.edt(t, anisotropy=an[::-1], black_border=bb).cpu().numpy().T
    assert np.array_equal(got, np.sqrt(want))


@pytest.mark.parametrize("shape,an,bb", [((1024, 1024, 200), (1.0, 1.0, 1.0), False), ((2048, 512, 300), (6.0, 6.0, 30.0), True),
                                         ((1024, 1024, 130), (4.0, 4.0, 40.0), False)])
def test_volumes_of_several_index_slabs_against_compiled_reference(edt_gpu, oracle_ref, shape, an, bb):
    """Round 6: volumes of more than 2^27 voxels run passes X and Y slab by slab (128 slices of 1024 x 1024) but keep EVERY slab's
    16-bit indices, so that the 16-bit plane between passes Y and Z exists there too (csrc/edt_api.hip: codes_whole).  Uneven
    last slabs (200 = 128 + 72, 300 = 128 + 128 + 44, 130 = 128 + 2), rows of 2048 voxels (two waves per row in pass X), both
    border rules, large cells next to small ones -- against the compiled reference, and against the same library with the
    single index slab of round 5 (EDT_HIP_WHOLE_INDEX_BYTES=0 in a subprocess: fp32 between Y and Z)."""
    import subprocess
    import sys
    import torch
    from edt import device

    lab = several_slabs_volume(shape)
    want = _ref_edtsq(oracle_ref, lab, an, bb)
    t = torch.from_numpy(np.ascontiguousarray(lab.T).view(np.int32)).cuda()
    got = device.edtsq(t, anisotropy=an[::-1], black_border=bb).cpu().numpy().T
    assert np.array_equal(got, want)
    del t, got, lab
    torch.cuda.empty_cache()
    # the same volume with one index slab (a fresh process: the limit is read once); compared through a digest of the result
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys, hashlib, numpy as np\n"
            f"sys.path[:0] = [{os.path.join(os.path.dirname(here), 'euclidean-distance-transform-3d_amd')!r}, {here!r}]\n"
            "import edt\n"
            "from synth import several_slabs_volume\n"
            f"out = edt.edtsq(several_slabs_volume({shape!r}), anisotropy={an!r}, black_border={bb!r})\n"
            "print('DIGEST', hashlib.sha256(np.ascontiguousarray(out.T).tobytes()).hexdigest())\n")
    env = dict(os.environ, EDT_HIP_WHOLE_INDEX_BYTES="0")
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "DIGEST " + hashlib.sha256(np.ascontiguousarray(want.T).tobytes()).hexdigest() in res.stdout, res.stdout[-300:]


def test_cfg4_1024_single_gpu_against_compiled_reference(edt_gpu, oracle_ref):
    import torch
    from edt import device

    lab, an, bb = config_volume("cfg4", 1024)
    want = _ref_edtsq(oracle_ref, lab, an, bb)
    t = torch.from_numpy(np.ascontiguousarray(lab.T).view(np.int32)).cuda()
    del lab
    got = device.edtsq(t, anisotropy=an[::-1], black_border=bb)
    del t
    assert np.array_equal(got.cpu().numpy().T, want)


@pytest.mark.parametrize("an,bb", [((6.0, 6.0, 30.0), True), ((1.0, 1.0, 1.0), False)])
def test_cfg5_512_sdf_against_compiled_reference(edt_gpu, oracle_ref, an, bb):
    """BASELINE configs[4], "... and sdf on 1xMI355X": sdf = edt(x) - edt(x == 0) (src/edt.pyx:121-158) of the 512^3 uint8
    blob volume, against the compiled reference run with every host thread -- bit for bit, both entry points."""
    import torch
    from edt import device

    lab, _, _ = config_volume("cfg5", 512)
    p = os.cpu_count() or 1
    want = oracle_ref.sdf(lab, an, bb, parallel=p)
    assert float(want.min()) < 0.0 < float(want.max())
    t = torch.from_numpy(np.ascontiguousarray(lab.T)).cuda()
    got = device.sdf(t, anisotropy=an[::-1], black_border=bb).cpu().numpy().T      # ONE transform (EDT_FLAG_SIGNED)
    assert np.array_equal(got, want)
    got = device._signed(t, an[::-1], bb, sqrt=True, one_transform=False).cpu().numpy().T   # the two-transform composition
    assert np.array_equal(got, want)
    del got
    assert np.array_equal(edt_gpu.sdf(lab, anisotropy=an, black_border=bb), want)
    if bb:
        want = oracle_ref.sdfsq(lab, an, bb, parallel=p)
        assert np.array_equal(edt_gpu.sdfsq(lab, anisotropy=an, black_border=bb), want)


def test_uneven_1024_world8_virtual_ranks_against_compiled_reference(edt_gpu, oracle_ref):
    """sz % 8 != 0 and sy % 8 != 0 at cfg4 size: ceil-sized leading slabs, y cut at multiples of 32 rows."""
    import torch
    from edt import _lib
    from edt.distributed import HipOps, balanced_partition

    shape = (1024, 1000, 1016)
    sx, sy, sz = shape
    world, chunks = 8, 4
    rng = np.random.default_rng(8)
    lab = np.asfortranarray(blocky_labels(shape, nlabels=60, zero_frac=0.03, block=24, rng=rng).astype(np.uint32))
    an, bb = (1.0, 1.0, 1.0), False
    want = _ref_edtsq(oracle_ref, lab, an, bb)
    dev = torch.device("cuda", 0)
    ops = HipOps()
    assert ops.records_supported(_lib.U32, sx, sy, sz)
    t = torch.from_numpy(np.ascontiguousarray(lab.T).view(np.int32)).to(dev)  # (sz, sy, sx), x fastest
    del lab
    words = -(-sy // 32)
    zparts = balanced_partition(sz, world)
    yparts = [(32 * a, min(32 * b, sy)) for a, b in balanced_partition(words, world)]
    y_splits = [a for a, _ in yparts] + [sy]
    rec = [ops.record_floats(sx, b - a) for a, b in yparts]
    dst = [torch.full((sz, rec[h]), float("nan"), dtype=torch.float32, device=dev) for h in range(world)]
    for r, (zs, ze) in enumerate(zparts):
        halo = t[zs - 1] if r > 0 else None  # the previous rank's last slice
        for c0, c1 in balanced_partition(ze - zs, chunks):
            blocks = [dst[h][zs + c0:zs + c1] if h == r else
                      torch.empty((c1 - c0, rec[h]), dtype=torch.float32, device=dev) for h in range(world)]
            ops.xy_records(t[zs + c0:zs + c1], halo, _lib.U32, an, 0, y_splits, blocks)
            for h in range(world):
                if h != r:
                    dst[h][zs + c0:zs + c1].copy_(blocks[h])  # the exchange
            halo = t[zs + c1 - 1]
    got = np.empty((sz, sy, sx), dtype=np.float32)
    for h, (ys, ye) in enumerate(yparts):
        ops.z_records(dst[h], sx, ye - ys, an[2], 0, wxy=(an[0], an[1]))
        got[:, ys:ye, :] = dst[h][:, :(ye - ys) * sx].reshape(sz, ye - ys, sx).cpu().numpy()
    assert np.array_equal(got.T, want)
    del dst
    # the same ranks with records of 16-bit rows (2.25 bytes per voxel: what an 8-GPU run of this volume exchanges)
    assert ops.records16_supported(_lib.U32, sx, sy, sz, an)
    rec16 = [ops.record16_words(sx, b - a) for a, b in yparts]
    dst16 = [torch.full((sz, rec16[h]), -1, dtype=torch.int32, device=dev) for h in range(world)]
    refused = torch.zeros(1, dtype=torch.int32, device=dev)
    for r, (zs, ze) in enumerate(zparts):
        halo = t[zs - 1] if r > 0 else None
        for c0, c1 in balanced_partition(ze - zs, chunks):
            blocks = [dst16[h][zs + c0:zs + c1] if h == r else
                      torch.empty((c1 - c0, rec16[h]), dtype=torch.int32, device=dev) for h in range(world)]
            ops.xy_records16(t[zs + c0:zs + c1], halo, _lib.U32, an, 0, y_splits, blocks, refused)
            for h in range(world):
                if h != r:
                    dst16[h][zs + c0:zs + c1].copy_(blocks[h])
            halo = t[zs + c1 - 1]
    del t
    assert int(refused.item()) == 0
    for h, (ys, ye) in enumerate(yparts):
        out = torch.empty((sz, ye - ys, sx), dtype=torch.float32, device=dev)
        ops.z_records16(dst16[h], out, an, 0)
        got[:, ys:ye, :] = out.cpu().numpy()
    assert np.array_equal(got.T, want)


def test_two_threads_two_streams_stay_bit_exact(edt_gpu, oracle_port):
    """Two host threads enqueue transforms on their own streams with their own plans at the same time (what a
    C++ host driving several GPUs, or several streams of one, does): no process-wide lock, no shared scratch."""
    import torch
    from edt import device

    dev = torch.device("cuda", 0)
    cases = []
    for seed, shape, an, bb in ((1, (160, 144, 96), (6.0, 6.0, 30.0), True), (2, (128, 200, 72), (1.0, 1.0, 1.0), False)):
        lab = voronoi_labels(shape, nseeds=50, seed=seed, upsample=4, membrane=0.03)
        cases.append((lab, an, bb, oracle_port.edtsq(lab, an, bb)))
    errors = []

    def worker(idx):
        try:
            lab, an, bb, want = cases[idx]
            torch.cuda.set_device(dev)
            stream = torch.cuda.Stream(dev)
            with torch.cuda.stream(stream):
                t = torch.from_numpy(np.ascontiguousarray(lab.T).view(np.int32)).to(dev, non_blocking=False)
                plan = device.Plan(lab.shape, 2, dev)
                out = torch.empty(t.shape, dtype=torch.float32, device=dev)
                for _ in range(40):
                    plan.run(t, an, black_border=bb, out=out)
                stream.synchronize()
                if not np.array_equal(out.cpu().numpy().T, want):
                    errors.append(f"thread {idx}: output differs from the oracle")
        except Exception as e:  # pragma: no cover
            errors.append(f"thread {idx}: {e!r}")

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_volume_beyond_2_32_voxels(edt_gpu):
    """A volume of more than 2^32 voxels (1280 x 2048 x 1664 = 4.36e9: 4.4 GB of uint8 labels, 17.4 GB of results) --
    what a 288 GB device is for; every element offset past 2^32 has to be 64-bit arithmetic in every kernel.  Labels: boxes
    of 160 x 256 x 208 voxels whose face neighbours all carry another label, black border: the squared distance of a voxel
    is the smallest of its three squared distances to the faces of its box (exact small integers times the voxel sizes),
    formed on the device by broadcasting.  Built and checked on the device; skipped where 60 GB are not free."""
    import torch
    from edt import _lib, device

    if torch.cuda.mem_get_info()[0] < 60 * (1 << 30):
        pytest.skip("needs 60 GB of free device memory")
    sx, sy, sz = 1280, 2048, 1664
    box = (160, 256, 208)
    w = (1.0, 2.0, 3.0)
    dev = torch.device("cuda", torch.cuda.current_device())
    idx = [torch.arange(s, device=dev) for s in (sx, sy, sz)]
    cell = [(i // b).to(torch.uint8) for i, b in zip(idx, box)]
    lab = ((cell[2][:, None, None] + cell[1][None, :, None] + cell[0][None, None, :]) % 3 + 1).contiguous()  # [z][y][x]
    assert lab.dtype == torch.uint8 and lab.numel() > 1 << 32
    face = [(torch.minimum(i % b + 1, b - i % b).to(torch.float32) * wa) ** 2 for i, b, wa in zip(idx, box, w)]
    plan = device.Plan((sx, sy, sz), _lib.U8, dev)
    out = plan.run(lab, w, black_border=True)
    torch.cuda.synchronize()
    del lab
    # compared slab by slab (no second 17 GB tensor)
    bad = 0
    for z0 in range(0, sz, 64):
        want = torch.minimum(torch.minimum(face[2][z0:z0 + 64, None, None], face[1][None, :, None]), face[0][None, None, :])
        bad += int((out[z0:z0 + 64] != want).sum())
    assert bad == 0
    # ... and the fused square root on the same plan, last slab only (the far end of every offset)
    out = plan.run(((cell[2][:, None, None] + cell[1][None, :, None] + cell[0][None, None, :]) % 3 + 1).contiguous(), w,
                   black_border=True, sqrt=True, out=out)
    torch.cuda.synchronize()
    want = torch.minimum(torch.minimum(face[2][-64:, None, None], face[1][None, :, None]), face[0][None, None, :]).sqrt()
    assert torch.equal(out[-64:], want)
    del out, plan
    torch.cuda.empty_cache()
