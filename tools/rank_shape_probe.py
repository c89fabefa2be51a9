#!/usr/bin/env python3
"""Per-phase kernel times of ONE rank of the 8-GPU run (BASELINE configs[3]: 1024^3 over 8 GPUs -> Z-slab
1024 x 1024 x 128, Y-slab 1024 x 128 x 1024), on one GPU: the XY phase writing slab records for 8 destinations,
the Z phase over the gathered records.  Labels: the cfg4 segmentation (bench.py: slab_labels)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "euclidean-distance-transform-3d_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import bench
from edt import _lib, device
from edt.distributed import HipOps, balanced_partition
dev = torch.device("cuda", 0)
world, rank = 8, 3
ext = (1024, 1024, 1024)
sx, sy, sz = ext
zparts = balanced_partition(sz, world)
words = sy // 32
yparts = [(32 * a, 32 * b) for a, b in balanced_partition(words, world)]
zs, ze = zparts[rank]
labels = bench.slab_labels(ext, zs, ze, dev, "cfg4")
halo = bench.slab_labels(ext, zs - 1, zs, dev, "cfg4")[0]
ops = HipOps()
rec = [ops.record_floats(sx, b - a) for a, b in yparts]
y_splits = [a for a, _ in yparts] + [sy]
dst = torch.zeros((sz, rec[rank]), dtype=torch.float32, device=dev)
blocks = [dst[zs:ze] if h == rank else torch.empty((ze - zs, rec[h]), dtype=torch.float32, device=dev) for h in range(world)]
an = (1.0, 1.0, 1.0)
def run():
    ops.xy_records(labels, halo, _lib.U32, an, 0, y_splits, blocks)
    ops.z_records(dst, sx, yparts[rank][1] - yparts[rank][0], an[2], 0, wxy=(an[0], an[1]))
for _ in range(2): run()
torch.cuda.synchronize()
device.set_profiling(True); acc = {}
for _ in range(5):
    run(); torch.cuda.synchronize()
    for k, v in device.pass_times(): acc.setdefault(k, []).append(v)
    device.set_profiling(False); device.set_profiling(True)
device.set_profiling(False)
t = {k: round(float(np.mean(v)), 3) for k, v in acc.items()}
print("rank-shape phases (ms):", t, "sum", round(sum(t.values()), 3), "(the Z phase here runs on records whose other slabs are zeros)")
t0 = time.perf_counter()
for _ in range(10): run()
torch.cuda.synchronize()
print("per step (both phases, no exchange): %.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
# the same rank with records of 16-bit rows (2.25 bytes per voxel; the Z phase writes a dense array)
if ops.records16_supported(_lib.U32, sx, sy, sz, an):
    rec16 = [ops.record16_words(sx, b - a) for a, b in yparts]
    dst16 = torch.zeros((sz, rec16[rank]), dtype=torch.int32, device=dev)
    blocks16 = [dst16[zs:ze] if h == rank else torch.empty((ze - zs, rec16[h]), dtype=torch.int32, device=dev) for h in range(world)]
    refused = torch.zeros(1, dtype=torch.int32, device=dev)
    out16 = torch.empty((sz, yparts[rank][1] - yparts[rank][0], sx), dtype=torch.float32, device=dev)
    def run16():
        ops.xy_records16(labels, halo, _lib.U32, an, 0, y_splits, blocks16, refused)
        ops.z_records16(dst16, out16, an, 0)
    for _ in range(2): run16()
    torch.cuda.synchronize()
    device.set_profiling(True); acc = {}
    for _ in range(5):
        run16(); torch.cuda.synchronize()
        for k, v in device.pass_times(): acc.setdefault(k, []).append(v)
        device.set_profiling(False); device.set_profiling(True)
    device.set_profiling(False)
    t = {k: round(float(np.mean(v)), 3) for k, v in acc.items()}
    t0 = time.perf_counter()
    for _ in range(10): run16()
    torch.cuda.synchronize()
    print("16-bit records, phases (ms):", t, "sum", round(sum(t.values()), 3), "| per step: %.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3),
          "| tiles without a 16-bit form:", int(refused.item()), "| bytes per record:", 4 * rec16[rank], "vs", 4 * rec[rank])
# the same slab through the single-device path (in-place Y pass instead of the scattering one)
out = torch.empty((ze - zs, sy, sx), dtype=torch.float32, device=dev)
plan = device.Plan((sx, sy, ze - zs), _lib.U32, dev)
for _ in range(2): plan.run(labels, an, black_border=False, out=out)
torch.cuda.synchronize()
device.set_profiling(True); acc = {}
for _ in range(5):
    plan.run(labels, an, black_border=False, out=out); torch.cuda.synchronize()
    for k, v in device.pass_times(): acc.setdefault(k, []).append(v)
device.set_profiling(False)
print("same slab, single-device path (ms):", {k: round(float(np.mean(v)), 3) for k, v in acc.items()})
