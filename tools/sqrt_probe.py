"""edt (fused correctly rounded sqrt in the last pass) against edtsq, device-resident, per configuration."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "euclidean-distance-transform-3d_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from edt import _lib, device
from synth import config_volume
lib = _lib.load()
for cfg in ("cfg2", "cfg3", "cfg3m"):
    lab_np, an, bb = config_volume(cfg, 512)
    lab = torch.from_numpy(np.ascontiguousarray(lab_np.T).view(np.int32)).cuda()
    out = torch.empty((512, 512, 512), dtype=torch.float32, device="cuda")
    plan = device.Plan((512, 512, 512), 2)
    res = {}
    for sq in (False, True):
        for _ in range(5): plan.run(lab, an, black_border=bb, sqrt=sq, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): plan.run(lab, an, black_border=bb, sqrt=sq, out=out)
        torch.cuda.synchronize(); res[sq] = (time.perf_counter() - t0) / 100 * 1e3
    print(f"{cfg}: edtsq {res[False]:.4f} ms   edt {res[True]:.4f} ms   (+{res[True] - res[False]:.4f})")
