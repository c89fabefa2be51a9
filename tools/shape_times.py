#!/usr/bin/env python3
"""Per-kernel hipEvent times for an arbitrary (sx, sy, sz) all-ones uint32 volume under debug modes.
usage: python tools/shape_times.py sx sy sz [mode ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "euclidean-distance-transform-3d_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from edt import _lib, device
from synth import box_edtsq_closed_form
lib = _lib.load()
sx, sy, sz = (int(v) for v in sys.argv[1:4])
modes = [int(m) for m in sys.argv[4:]] or [0]
dev = torch.device("cuda", 0)
lab = torch.ones((sz, sy, sx), dtype=torch.int32, device=dev)
out = torch.empty((sz, sy, sx), dtype=torch.float32, device=dev)
plan = device.Plan((sx, sy, sz), 2, dev)
an = (6.0, 6.0, 30.0)
want = None
for mode in modes:
    lib.edt_hip_set_debug_mode(mode)
    device.set_profiling(True)
    acc = {}
    for _ in range(6):
        plan.run(lab, an, black_border=True, out=out); torch.cuda.synchronize()
        for k, v in device.pass_times(): acc.setdefault(k, []).append(v)
    device.set_profiling(False)
    if want is None:
        want = box_edtsq_closed_form((sx, sy, sz), an)
    ok = bool(np.array_equal(out.cpu().numpy().T, want))
    print("debug_mode", mode, {k: round(float(np.mean(v[1:])), 4) for k, v in acc.items()}, "verified" if ok else "WRONG")
lib.edt_hip_set_debug_mode(0)
