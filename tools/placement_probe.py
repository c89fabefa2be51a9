"""Does the speed of the kernels depend on WHERE the loader puts their code?  (diagnostics)
A device allocation of PAD_KB KiB made before the library's first kernel launch shifts everything the
runtime allocates afterwards, including the code objects it loads lazily.  One process per value:
    for p in 0 4 64 ...; do PAD_KB=$p python tools/placement_probe.py; done"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "euclidean-distance-transform-3d_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
pad_kb = int(os.environ.get("PAD_KB", "0"))
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
if pad_kb:
    p = ctypes.c_void_p()
    rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(pad_kb * 1024))
    assert rc == 0
from edt import _lib, device
n = 512
lab = torch.ones((n, n, n), dtype=torch.int32, device=dev)
out = torch.empty((n, n, n), dtype=torch.float32, device=dev)
plan = device.Plan((n, n, n), 2, dev)
device.set_profiling(True)
acc = {}
for _ in range(8):
    plan.run(lab, (6.0, 6.0, 30.0), black_border=True, out=out); torch.cuda.synchronize()
    for k, v in device.pass_times(): acc.setdefault(k, []).append(v)
print("pad_kb", pad_kb, {k: round(float(np.mean(v[2:])), 4) for k, v in acc.items()})
