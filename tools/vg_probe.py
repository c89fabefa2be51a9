"""cfg5 voxel-graph transform, device-resident: 512^3 uint8 + uint8 graph -> fp32 (ms per call)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "euclidean-distance-transform-3d_amd"))
import numpy as np, torch
from edt import device
rng = np.random.default_rng(5)
n = 512
small = (rng.random((32, 32, 32)) < 0.6).astype(np.uint8)
lab = torch.from_numpy(np.ascontiguousarray(small.repeat(16, 0).repeat(16, 1).repeat(16, 2))).cuda()
g = torch.full((n, n, n), 0b00111111, dtype=torch.uint8, device="cuda")
g[torch.rand((n, n, n), device="cuda") < 0.01] = 0b00111110
for _ in range(2):
    out = device.edtsq_voxel_graph(lab, g, anisotropy=(30.0, 6.0, 6.0), black_border=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    out = device.edtsq_voxel_graph(lab, g, anisotropy=(30.0, 6.0, 6.0), black_border=True)
torch.cuda.synchronize()
print(f"device-resident voxel_graph edtsq 512^3: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per call")
