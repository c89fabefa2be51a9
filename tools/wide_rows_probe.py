#!/usr/bin/env python3
"""Pass X on rows of 1025..2048 voxels: the two-waves-per-row form (edt_rowwave.hip, H = 2) against the workgroup-phased
kernel (debug bit 0x4000000), 2^27 voxels each.  usage: python tools/wide_rows_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "euclidean-distance-transform-3d_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from edt import _lib, device
from synth import blocky_labels
lib = _lib.load()
dev = torch.device("cuda", 0)
for shape in ((2048, 2048, 32), (1280, 1024, 102), (1536, 512, 170), (1100, 1100, 110)):
    sx, sy, sz = shape
    rng = np.random.default_rng(1)
    lab_np = blocky_labels((sx // 8 + 1, sy // 8 + 1, sz // 8 + 1), nlabels=50, zero_frac=0.05, block=1, rng=rng).astype(np.uint32)
    lab_np = np.kron(lab_np, np.ones((8, 8, 8), dtype=np.uint32))[:sx, :sy, :sz]
    lab = torch.from_numpy(np.ascontiguousarray(lab_np.T).view(np.int32)).to(dev)
    plan = device.Plan(shape, _lib.U32, dev)
    outs = []
    for mode in (0, 0x4000000):
        lib.edt_hip_set_debug_mode(mode)
        out = torch.empty((sz, sy, sx), dtype=torch.float32, device=dev)
        device.set_profiling(True); acc = {}
        for _ in range(6):
            plan.run(lab, (1.0, 1.0, 1.0), black_border=False, out=out); torch.cuda.synchronize()
            for k, v in device.pass_times(): acc.setdefault(k, []).append(v)
        device.set_profiling(False)
        lib.edt_hip_set_debug_mode(0)
        t = {k: round(float(np.mean(v[1:])), 3) for k, v in acc.items()}
        outs.append(out)
        print(f"{shape} mode {mode:#x}: {t} total {sum(t.values()):.3f} ms")
    print("  same bits:", bool(torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))))
