import sys, os
sys.path.insert(0, "euclidean-distance-transform-3d_amd"); sys.path.insert(0, "tests")
import numpy as np
import edt
from edt import _lib
lib = _lib.load()
lib.edt_hip_set_debug_mode(0x1000)
lab = np.ones((64, 48), dtype=np.uint32, order="F")
print("2d", flush=True)
r = edt.edtsq(lab, anisotropy=(1, 1), black_border=True)
print(r.max(), flush=True)
lab = np.ones((64, 48, 40), dtype=np.uint32, order="F")
print("3d", flush=True)
r = edt.edtsq(lab, anisotropy=(1, 1, 1), black_border=True)
print(r.max(), flush=True)
