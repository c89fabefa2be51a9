#!/usr/bin/env python3
"""Per-kernel times on a full-resolution (not up-sampled) Voronoi segmentation: smooth label boundaries,
the closest synthetic stand-in for an EM segmentation.  usage: python tools/voronoi_times.py [n] [nseeds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "euclidean-distance-transform-3d_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from edt import _lib, device
from synth import voronoi_labels
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 250
lib = _lib.load()
dev = torch.device("cuda", 0)
for up, membrane in ((1, 0.0), (1, 0.02), (4, 0.0)):
    lab_np = voronoi_labels((n, n, n), nseeds, seed=3, upsample=up, membrane=membrane)
    lab = torch.from_numpy(np.ascontiguousarray(lab_np.T).view(np.int32)).to(dev)
    out = torch.empty((n, n, n), dtype=torch.float32, device=dev)
    plan = device.Plan((n, n, n), 2, dev)
    for an, bb in (((1.0, 1.0, 1.0), False), ((6.0, 6.0, 30.0), False)):
        device.set_profiling(True); acc = {}
        for _ in range(6):
            plan.run(lab, an, black_border=bb, out=out); torch.cuda.synchronize()
            for k, v in device.pass_times(): acc.setdefault(k, []).append(v)
        device.set_profiling(False)
        t = {k: round(float(np.mean(v[1:])), 4) for k, v in acc.items()}
        tot = sum(t.values())
        print(f"upsample={up} membrane={membrane} anis={an}: {t} total {tot:.3f} ms = {n**3/tot/1e3:.0f} Mvox/s")
