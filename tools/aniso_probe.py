#!/usr/bin/env python3
"""Per-pass times of one 512^3 configuration under voxel sizes with exact (fp32 candidates, 16-bit indices between X and
Y) and inexact multiples (fp64 candidates, fp32 between X and Y).  usage: python tools/aniso_probe.py [cfg3|cfg3M|cfg3L]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "euclidean-distance-transform-3d_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from edt import device, _lib
from synth import config_volume
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
lab_np, _, bb = config_volume(cfg, 512)
dev = torch.device("cuda", 0)
lab = torch.from_numpy(np.ascontiguousarray(lab_np.T).view(np.int32)).to(dev)
out = torch.empty((512, 512, 512), dtype=torch.float32, device=dev)
plan = device.Plan((512, 512, 512), 2, dev)
for an in ((1.0, 1.0, 1.0), (4.0, 4.0, 40.0), (6.0, 6.0, 30.0), (3.58, 3.58, 40.0), (0.7, 1.3, 2.1), (1.1, 1.1, 1.1)):
    device.set_profiling(True); acc = {}
    for _ in range(6):
        plan.run(lab, an, black_border=bb, out=out); torch.cuda.synchronize()
        for k, v in device.pass_times(): acc.setdefault(k, []).append(v)
    device.set_profiling(False)
    t = {k: round(float(np.mean(v[1:])), 3) for k, v in acc.items()}
    # the same call with every tile on the hull path (0x2000) and with fp64 candidates on the windowed path (0x8000 |
    # 0x2000000): three forms of one computation, compared bit for bit on the device
    same = []
    for mode in (0x2000, 0x8000 | 0x2000000):
        _lib.load().edt_hip_set_debug_mode(mode)
        other = torch.empty_like(out)
        plan.run(lab, an, black_border=bb, out=other); torch.cuda.synchronize()
        _lib.load().edt_hip_set_debug_mode(0)
        same.append(bool(torch.equal(out.view(torch.int32), other.view(torch.int32))))
    print(f"{cfg} anisotropy {an}: {t} total {sum(t.values()):.3f} ms; same bits as hull form / fp64 candidates: {same}")
