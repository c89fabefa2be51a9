#!/usr/bin/env python3
"""Calibrate the box: achievable HBM copy bandwidth (torch copy) and the column pass's memory-only floor."""
import sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "euclidean-distance-transform-3d_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from edt import _lib, device
lib = _lib.load()
dev = torch.device("cuda", 0)
n = 512
a = torch.rand(n * n * n, device=dev); b = torch.empty_like(a)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
t = timeit(lambda: b.copy_(a)); print(f"torch copy 512MB: {t*1e3:.3f} ms  {2*a.numel()*4/t/1e12:.2f} TB/s (r+w)")
a4 = torch.rand(4 * n * n * n, device=dev); b4 = torch.empty_like(a4)
t = timeit(lambda: b4.copy_(a4)); print(f"torch copy 2GB: {t*1e3:.3f} ms  {2*a4.numel()*4/t/1e12:.2f} TB/s (r+w)")
t = timeit(lambda: torch.add(a, 1.0, out=b)); print(f"torch add 512MB: {t*1e3:.3f} ms  {2*a.numel()*4/t/1e12:.2f} TB/s")
t = timeit(lambda: a.sum()); print(f"torch sum 512MB (read only): {t*1e3:.3f} ms  {a.numel()*4/t/1e12:.2f} TB/s")
t = timeit(lambda: b.fill_(1.0)); print(f"torch fill 512MB (write only): {t*1e3:.3f} ms  {a.numel()*4/t/1e12:.2f} TB/s")
from synth import config_volume
cfg = os.environ.get("MB_CFG", "cfg2")
lab_np, an, bb = config_volume(cfg, n)
lab = torch.from_numpy(np.ascontiguousarray(lab_np.T).view(np.int32)).to(dev); out = torch.empty((n, n, n), dtype=torch.float32, device=dev)
plan = device.Plan((n, n, n), 2, dev)
for mode in [int(m, 0) for m in os.environ.get('MB_MODES', '0,14').split(',')]:
    lib.edt_hip_set_debug_mode(mode)
    device.set_profiling(True)
    acc = {}
    for _ in range(5):
        plan.run(lab, an, black_border=bb, out=out); torch.cuda.synchronize()
        for k, v in device.pass_times(): acc.setdefault(k, []).append(v)
    device.set_profiling(False)
    print("debug_mode", mode, {k: round(float(np.mean(v[1:])), 4) for k, v in acc.items()})
lib.edt_hip_set_debug_mode(0)
