#!/usr/bin/env python3
"""Do the passes care WHERE their buffers lie relative to one another?  The workspace (bit planes + the 16-bit index / plane buffer) and
the output are placed at a grid of offsets inside over-sized allocations and the per-pass times of cfg2 / cfg3 taken for each.
usage: python tools/placement_sweep.py [cfg] [step_KiB] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "euclidean-distance-transform-3d_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from edt import device
from synth import config_volume
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
step = int(sys.argv[2]) * 1024 if len(sys.argv) > 2 else 256 * 1024
count = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda", 0)
lab_np, an, bb = config_volume(cfg, 512)
lab = torch.from_numpy(np.ascontiguousarray(lab_np.T).view(np.int32)).to(dev)
plan = device.Plan(lab_np.shape, 2, dev)
nws = plan.workspace.numel()
slack = step * count
big_ws = torch.empty(nws + slack, dtype=torch.uint8, device=dev)
big_out = torch.empty(lab.numel() * 4 + slack, dtype=torch.uint8, device=dev)
print("labels %x ws %x out %x" % (lab.data_ptr(), big_ws.data_ptr(), big_out.data_ptr()))
res = {}
for i in range(count):
    for j in range(count):
        plan.workspace = big_ws[i * step: i * step + nws]
        out = big_out[j * step: j * step + lab.numel() * 4].view(torch.float32).view(lab.shape)
        for _ in range(6):
            plan.run(lab, an, black_border=bb, out=out)
        torch.cuda.synchronize()
        device.set_profiling(True)
        acc = {}
        for _ in range(8):
            plan.run(lab, an, black_border=bb, out=out); torch.cuda.synchronize()
            for k, v in device.pass_times(): acc.setdefault(k, []).append(v)
        device.set_profiling(False)
        t = {k: float(np.mean(v)) for k, v in acc.items()}
        res[(i, j)] = t
        print("ws+%4dK out+%4dK  x %.4f y %.4f z %.4f  sum %.4f" % (i * step // 1024, j * step // 1024, t["x_pass"], t["y_pass"], t["z_pass"], sum(t.values())), flush=True)
zs = np.array([[res[(i, j)]["z_pass"] for j in range(count)] for i in range(count)])
ys = np.array([[res[(i, j)]["y_pass"] for j in range(count)] for i in range(count)])
xs = np.array([[res[(i, j)]["x_pass"] for j in range(count)] for i in range(count)])
for name, m in (("x", xs), ("y", ys), ("z", zs)):
    print(name, "min %.4f max %.4f  by ws shift" % (m.min(), m.max()), np.round(m.mean(1), 4).tolist(), " by out shift", np.round(m.mean(0), 4).tolist())
