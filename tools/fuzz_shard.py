#!/usr/bin/env python3
"""Randomised parity of the two sharded phases as VIRTUAL ranks on one GPU (edt_hip_shard_xy_records[16]_device /
edt_hip_shard_z_records[16]_device; the exchange is a copy): random extents, world sizes, chunk counts, voxel sizes, border
modes, label structures -- records of 16-bit rows where the library says they apply (a case whose XY phase counts a tile
without a 16-bit form is repeated with fp32 rows, as the driver does), fp32 rows otherwise.  GPU vs oracle, bit for bit.
usage: python tools/fuzz_shard.py [ncases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "euclidean-distance-transform-3d_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from edt import _lib
from edt.distributed import HipOps, balanced_partition
from oracle import harness
from synth import blocky_labels
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
if not harness.have_port():
    harness.build("port")
o = harness.port()
dev = torch.device("cuda", 0)
ops = HipOps()
bad = used16 = fell = 0
t0 = time.time()
for i in range(ncases):
    sx = 4 * int(rng.integers(2, 80)) if rng.random() < 0.8 else int(rng.integers(8, 300))
    sy = int(rng.integers(64, 420))
    sz = int(rng.integers(8, 360))
    if sx * sy * sz > 2.5e7:
        continue
    words = -(-sy // 32)
    world = int(rng.integers(1, min(8, words, sz) + 1))
    chunks = int(rng.integers(1, 5))
    an = tuple(float(a) for a in rng.choice([1, 2, 6, 30, 4, 3, 0.5, 1.3], size=3))
    bb = bool(rng.integers(0, 2))
    sqrt = rng.random() < 0.3
    kind = rng.integers(0, 4)
    if kind == 0:
        lab = np.ones((sx, sy, sz), dtype=np.uint32)
    else:
        lab = blocky_labels((sx, sy, sz), nlabels=int(rng.integers(1, 60)), zero_frac=float(rng.random() * 0.2),
                            block=int(rng.integers(2, 90)), rng=rng).astype(np.uint32)
    lab = np.asfortranarray(lab)
    want = o.edtsq(lab, an, bb)
    if sqrt:
        want = np.sqrt(want)
    t = torch.from_numpy(np.ascontiguousarray(lab.T).view(np.int32)).to(dev)
    zparts = balanced_partition(sz, world)
    yparts = [(32 * a, min(32 * b, sy)) for a, b in balanced_partition(words, world)]
    y_splits = [a for a, _ in yparts] + [sy]
    flags = _lib.FLAG_BLACK_BORDER if bb else 0
    zflags = flags | (_lib.FLAG_SQRT if sqrt else 0)
    if not ops.records_supported(_lib.U32, sx, sy, sz):
        continue
    use16 = ops.records16_supported(_lib.U32, sx, sy, sz, an)
    got = None
    for attempt in (16, 32):
        if attempt == 16 and not use16:
            continue
        if attempt == 16:
            rec = [ops.record16_words(sx, b - a) for a, b in yparts]
            dst = [torch.full((sz, rec[h]), -1, dtype=torch.int32, device=dev) for h in range(world)]
            refused = torch.zeros(1, dtype=torch.int32, device=dev)
        else:
            rec = [ops.record_floats(sx, b - a) for a, b in yparts]
            dst = [torch.full((sz, rec[h]), float("nan"), dtype=torch.float32, device=dev) for h in range(world)]
        for r, (zs, ze) in enumerate(zparts):
            halo = t[zs - 1] if r > 0 else None
            for c0, c1 in balanced_partition(ze - zs, min(chunks, ze - zs)):
                blocks = [dst[h][zs + c0:zs + c1] if h == r else torch.empty((c1 - c0, rec[h]), dtype=dst[h].dtype, device=dev)
                          for h in range(world)]
                if attempt == 16:
                    ops.xy_records16(t[zs + c0:zs + c1], halo, _lib.U32, an, flags, y_splits, blocks, refused)
                else:
                    ops.xy_records(t[zs + c0:zs + c1], halo, _lib.U32, an, flags, y_splits, blocks)
                for h in range(world):
                    if h != r:
                        dst[h][zs + c0:zs + c1].copy_(blocks[h])
                halo = t[zs + c1 - 1]
        if attempt == 16 and int(refused.item()) != 0:
            fell += 1
            continue   # (tiles without a 16-bit form: the step is repeated with fp32 rows)
        outs = []
        for h, (ys, ye) in enumerate(yparts):
            if attempt == 16:
                out = torch.full((sz, ye - ys, sx), float("nan"), dtype=torch.float32, device=dev)
                ops.z_records16(dst[h], out, an, zflags)
            else:
                ops.z_records(dst[h], sx, ye - ys, an[2], zflags, wxy=(an[0], an[1]))
                out = dst[h][:, :(ye - ys) * sx].reshape(sz, ye - ys, sx)
            outs.append(out)
        got = torch.cat(outs, 1).cpu().numpy().T
        used16 += attempt == 16
        break
    if got is None or not np.array_equal(got, want, equal_nan=True):
        bad += 1
        print("MISMATCH", (sx, sy, sz), world, chunks, an, bb, sqrt, "16-bit" if use16 else "fp32")
print(f"{ncases} cases, {bad} mismatches, {used16} over 16-bit records ({fell} fell back to fp32 rows), {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
