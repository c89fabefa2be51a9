"""Small volumes are launch-bound: eager edt_hip_edtsq_device (4 kernel launches from Python) vs. the same
transform captured once into a hipGraph and replayed.  python tools/graph_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "euclidean-distance-transform-3d_amd"))
import torch
from edt import device
dev = torch.device("cuda", 0)
for n in (32, 64, 128, 256):
    t = torch.ones((n, n, n), dtype=torch.int32, device=dev)
    out = torch.empty((n, n, n), dtype=torch.float32, device=dev)
    plan = device.Plan((n, n, n), 2, dev)
    an = (6.0, 6.0, 30.0)
    for _ in range(3):
        plan.run(t, an, black_border=True, out=out)
    torch.cuda.synchronize()
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.run(t, an, black_border=True, out=out)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / reps * 1e6
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            plan.run(t, an, black_border=True, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / reps * 1e6
    print(f"{n}^3: eager {eager:.1f} us/transform, hipGraph replay {graph:.1f} us/transform")
