"""cfg5's sdf leg, device-resident: sdf = edt(x) - edt(x == 0) of the 512^3 uint8 blob volume as ONE transform (EDT_FLAG_SIGNED)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "euclidean-distance-transform-3d_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from edt import device
from synth import config_volume
lab_np, _, _ = config_volume("cfg5", 512)
lab = torch.from_numpy(np.ascontiguousarray(lab_np.T)).cuda()
for _ in range(3):
    out = device.sdf(lab, anisotropy=(30.0, 6.0, 6.0), black_border=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    out = device.sdf(lab, anisotropy=(30.0, 6.0, 6.0), black_border=True)
torch.cuda.synchronize()
print(f"device-resident sdf 512^3 uint8: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per call")
