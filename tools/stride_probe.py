#!/usr/bin/env python3
"""Per-pass hipEvent times of a segmentation-like volume (random labels in cells of 16 voxels, uint32, anisotropy
(6, 6, 30)) at shapes that differ only in their strides: does the Z pass of 1024^3 -- rows 2 / 4 MiB apart -- pay for the
power-of-two stride?   usage: python tools/stride_probe.py sx,sy,sz [sx,sy,sz ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "euclidean-distance-transform-3d_amd"))
import numpy as np, torch
from edt import device
dev = torch.device("cuda", 0)
for arg in sys.argv[1:]:
    sx, sy, sz = (int(v) for v in arg.split(","))
    g = torch.Generator(device=dev); g.manual_seed(1)
    coarse = torch.randint(1, 2000, ((sz + 15) // 16, (sy + 15) // 16, (sx + 15) // 16), device=dev, dtype=torch.int32, generator=g)
    lab = coarse.repeat_interleave(16, 0).repeat_interleave(16, 1).repeat_interleave(16, 2)[:sz, :sy, :sx].contiguous()
    out = torch.empty((sz, sy, sx), dtype=torch.float32, device=dev)
    plan = device.Plan((sx, sy, sz), 2, dev)
    device.set_profiling(True)
    acc = {}
    for _ in range(5):
        plan.run(lab, (6.0, 6.0, 30.0), black_border=True, out=out); torch.cuda.synchronize()
        for k, v in device.pass_times(): acc.setdefault(k, []).append(v)
    device.set_profiling(False)
    mv = sx * sy * sz / 1e6
    t = {k: float(np.mean(v[1:])) for k, v in acc.items()}
    print(arg, {k: round(v, 3) for k, v in t.items()}, "total %.3f ms" % sum(t.values()), "| ns/voxel:", {k: round(v * 1e6 / (mv * 1e6), 4) for k, v in t.items()})
    del lab, out, plan
