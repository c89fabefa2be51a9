import sys, time, os
sys.path.insert(0, 'euclidean-distance-transform-3d_amd'); sys.path.insert(0, 'tests')
import numpy as np, torch
from edt import device
from synth import config_volume
lab_np, an, bb = config_volume("cfg3", 512)
t = torch.from_numpy(np.ascontiguousarray(lab_np.T).view(np.int32)).cuda()
dt = device.edt(t, anisotropy=an[::-1], black_border=bb)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 0
for k, img in device.each(t, dt, in_place=True):
    n += 1
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"device.each over {n} labels of 512^3: {el:.3f} s total, {el / n * 1e3:.3f} ms per label ({2 * 4 * t.numel() / (el / n) / 1e12 + 4 * t.numel() / (el / n) / 1e12:.2f} TB/s)")
