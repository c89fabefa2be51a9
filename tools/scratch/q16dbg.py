import sys, os
sys.path.insert(0, "euclidean-distance-transform-3d_amd"); sys.path.insert(0, "tests")
import numpy as np
import edt
from edt import _lib
lib = _lib.load()
lib.edt_hip_set_debug_mode(0x1000)
lab = np.ones((128, 160, 136), dtype=np.uint32, order="F")
try:
    r = edt.edtsq(lab, anisotropy=(1, 1, 1), black_border=True)
    print("ok", r.max(), flush=True)
except Exception as e:
    print("ERR", e)
