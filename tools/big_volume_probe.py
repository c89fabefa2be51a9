#!/usr/bin/env python3
"""Time one volume of more than 2^32 voxels (1280 x 2048 x 1664 uint8, the boxes of tests/test_gpu_fullsize.py::
test_volume_beyond_2_32_voxels) on one device: per-pass times and Mvox/s, against the 512^3 rate.
usage: python tools/big_volume_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "euclidean-distance-transform-3d_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from edt import _lib, device

sx, sy, sz = 1280, 2048, 1664
box = (160, 256, 208)
dev = torch.device("cuda", 0)
idx = [torch.arange(s, device=dev) for s in (sx, sy, sz)]
cell = [(i // b).to(torch.uint8) for i, b in zip(idx, box)]
lab = ((cell[2][:, None, None] + cell[1][None, :, None] + cell[0][None, None, :]) % 3 + 1).contiguous()
plan = device.Plan((sx, sy, sz), _lib.U8, dev)
out = torch.empty((sz, sy, sx), dtype=torch.float32, device=dev)
for an in ((1.0, 2.0, 3.0), (1.1, 1.1, 1.1)):
    device.set_profiling(True); acc = {}
    for _ in range(4):
        plan.run(lab, an, black_border=True, out=out); torch.cuda.synchronize()
        for k, v in device.pass_times(): acc.setdefault(k, []).append(v)
    device.set_profiling(False)
    t = {k: round(float(np.mean(v[1:])), 2) for k, v in acc.items()}
    tot = sum(t.values())
    print(f"{sx}x{sy}x{sz} uint8 ({lab.numel() / 1e9:.2f} Gvoxel), boxes {box}, anisotropy {an}: {t} total {tot:.2f} ms = "
          f"{lab.numel() / tot / 1e3:.0f} Mvox/s; workspace {plan.workspace.numel() / 2**30:.2f} GiB")
