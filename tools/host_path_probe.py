#!/usr/bin/env python3
"""numpy in -> numpy out through the C ABI (edt_hip_edt3dsq), 512^3 uint32: where the wall time goes."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "euclidean-distance-transform-3d_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import edt
from edt import _lib
lib = _lib.load()
n = 512
lab = np.ones((n, n, n), dtype=np.uint32, order="F")
def best(fn, reps=4):
    fn(); b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); b = min(b, time.perf_counter() - t0)
    return b * 1e3
out = np.zeros(lab.size, dtype=np.float32); out[:] = 1  # resident pages
def abi():
    rc = lib.edt_hip_edt3dsq(lab.ctypes.data, _lib.U32, n, n, n, 6.0, 6.0, 30.0, 1, 1, out.ctypes.data)
    assert rc == 0
print("C ABI, result buffer reused (resident pages): %.1f ms" % best(abi))
keep = []
def fresh_keep():
    keep.append(edt.edtsq(lab, anisotropy=(6, 6, 30), black_border=True))
print("edt.edtsq, fresh result array each call (kept alive): %.1f ms" % best(fresh_keep))
keep.clear()
r = [None]
def fresh_drop():
    r[0] = edt.edtsq(lab, anisotropy=(6, 6, 30), black_border=True)
print("edt.edtsq, fresh result array, previous one freed: %.1f ms" % best(fresh_drop))
t0 = time.perf_counter(); a = np.empty(lab.size, np.float32); a[::1024] = 0; print("first touch of 512 MiB, one thread: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); del a; print("free of 512 MiB: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
