"""Per-pass times of the 512^3 all-ones volume for every label dtype (pass X is the only one that sees it)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "euclidean-distance-transform-3d_amd"))
import numpy as np, torch
from edt import _lib, device
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
out = torch.empty((n, n, n), dtype=torch.float32, device=dev)
for name, dt in (("uint8", torch.uint8), ("int16", torch.int16), ("int32", torch.int32), ("int64", torch.int64),
                 ("float32", torch.float32), ("float64", torch.float64), ("bool", torch.bool)):
    lab = torch.ones((n, n, n), dtype=dt, device=dev)
    plan = device.Plan((n, n, n), device.dtype_code(dt), dev)
    device.set_profiling(True)
    acc = {}
    for _ in range(6):
        plan.run(lab, (6.0, 6.0, 30.0), black_border=True, out=out); torch.cuda.synchronize()
        for k, v in device.pass_times(): acc.setdefault(k, []).append(v)
    device.set_profiling(False)
    ok = float(out.max()) == (6.0 * n / 2) ** 2
    print(f"{name:8s}", {k: round(float(np.mean(v[1:])), 4) for k, v in acc.items()}, "max ok" if ok else "MAX WRONG")
    del lab, plan
