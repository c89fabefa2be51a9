#!/usr/bin/env python3
"""Per-pass times of the single-device path on Z-slabs of the 1024^3 segmentation (1024 x 1024 x nz): does a pass
cost the same per voxel in a thin slab as in the cube?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "euclidean-distance-transform-3d_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import bench
from edt import _lib, device
dev = torch.device("cuda", 0)
ext = (1024, 1024, 1024)
full = bench.slab_labels(ext, 0, 1024, dev, "cfg4")
for nz in (32, 64, 128, 256, 512, 1024):
    lab = full[384:384 + nz] if nz < 1024 else full
    lab = lab.contiguous()
    out = torch.empty(lab.shape, dtype=torch.float32, device=dev)
    plan = device.Plan((1024, 1024, nz), _lib.U32, dev)
    for _ in range(2): plan.run(lab, (1.0, 1.0, 1.0), black_border=False, out=out)
    torch.cuda.synchronize()
    device.set_profiling(True); acc = {}
    for _ in range(4):
        plan.run(lab, (1.0, 1.0, 1.0), black_border=False, out=out); torch.cuda.synchronize()
        for k, v in device.pass_times(): acc.setdefault(k, []).append(v)
    device.set_profiling(False)
    units = nz / 128.0
    print(f"nz={nz:5d}: per 134M voxels: " + ", ".join(f"{k} {np.mean(v) / units:.3f}" for k, v in acc.items()))
    del plan, out
