#!/usr/bin/env python3
"""Windowed path vs hull path of the column kernel over segmentations of growing cell size (512^3, full-resolution
Voronoi cells: smooth boundaries): where does each form win?  usage: python tools/window_sweep.py [n]
Run once per mode: EDT_HIP_DEBUG_MODE=0 (per-tile choice), 0x2000 (hull only), 0x4000 (window on every tile)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "euclidean-distance-transform-3d_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from edt import device
from synth import voronoi_labels
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
SEEDS = [int(v) for v in os.environ.get('WINDOW_SWEEP_SEEDS', '8000,2000,500,120,30,8').split(',')]
for nseeds in SEEDS:
    lab_np = voronoi_labels((n, n, n), nseeds, seed=3, upsample=2, membrane=0.0)
    lab = torch.from_numpy(np.ascontiguousarray(lab_np.T).view(np.int32)).to(dev)
    out = torch.empty((n, n, n), dtype=torch.float32, device=dev)
    plan = device.Plan((n, n, n), 2, dev)
    for an, bb in (((1.0, 1.0, 1.0), False),):
        device.set_profiling(True); acc = {}
        for _ in range(5):
            plan.run(lab, an, black_border=bb, out=out); torch.cuda.synchronize()
            for k, v in device.pass_times(): acc.setdefault(k, []).append(v)
        device.set_profiling(False)
        t = {k: round(float(np.mean(v[1:])), 3) for k, v in acc.items()}
        print(f"mode={os.environ.get('EDT_HIP_DEBUG_MODE','0')} limit={os.environ.get('EDT_HIP_WINDOW_LIMIT','-')} seeds={nseeds:5d} (cell ~{n/nseeds**(1/3):.0f} vox) max edt {float(out[torch.isfinite(out)].max())**0.5:.0f}: y {t.get('y_pass')} z {t.get('z_pass')} total {sum(t.values()):.3f} ms")
