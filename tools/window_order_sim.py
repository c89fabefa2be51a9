#!/usr/bin/env python3
"""How many window steps does a WAVE of the windowed path execute per block, and what would other assignments of blocks
to wave iterations save?  Host model, no GPU: the Y pass of z-slices of a 512^3 configuration; a row needs
ceil(sqrt(result / w2)) steps (its window ends once c_d >= its minimum), a block of 8 rows the maximum of its rows, a wave
(32 adjacent columns x 2 bands, lane = column x band, four blocks per lane one after the other) the maximum over its 64
lanes in every one of its four iterations.  Compared (plus other shapes of the 64 blocks of one iteration): the kernel's order (block k of every lane in iteration k); every
lane's blocks in descending order of their windows; blocks of 4 rows; and the bound no assignment beats (every lane busy:
the mean).  usage: python tools/window_order_sim.py [cfg3|cfg3M|cfg3L] [slices]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
from oracle import harness
from synth import config_volume
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
nsl = int(sys.argv[2]) if len(sys.argv) > 2 else 6
lab, an, bb = config_volume(cfg, 512)
if not harness.have_port():
    harness.build("port")
o = harness.port()
tot = {"kernel order": 0.0, "sorted per lane": 0.0, "4-row blocks": 0.0, "mean (bound)": 0.0}
for z in np.linspace(0, 511, nsl).astype(int):
    sl = np.asfortranarray(lab[:, :, z])
    r = o.edtsq(sl, (an[0], an[1]), bb)                   # the Y pass's result on this slice, [x][y]
    W = np.ceil(np.sqrt(r / np.float32(an[1]) ** 2)).astype(np.int64)  # steps a row needs
    W = W.T                                               # [y][x]
    n, sx = W.shape
    Wb = W.reshape(n // 8, 8, sx).max(1)                  # [block][x]
    Wb4 = W.reshape(n // 4, 4, sx).max(1)
    # lanes of a wave: 32 adjacent columns x 2 adjacent bands; a band = 4 blocks
    B = Wb.reshape(n // 32, 4, sx)                        # [band][k][x]
    B = B.reshape(n // 64, 2, 4, sx // 32, 32).transpose(0, 3, 2, 1, 4).reshape(n // 64, sx // 32, 4, 64)  # [..][k][lane]
    tot["kernel order"] += B.max(-1).sum()
    tot["sorted per lane"] += np.sort(B, axis=2).max(-1).sum()
    tot["mean (bound)"] += B.mean(-1).sum()
    # other shapes of the 64 blocks a wave works on at a time (all of them 64 lanes x one block of 8 rows)
    for name, arr in (("32 cols x 16 contiguous rows", Wb.reshape(n // 64, 4, 2, sx // 32, 32).max(axis=(2, 4))),
                      ("16 cols x 32 contiguous rows", Wb.reshape(n // 128, 4, 4, sx // 16, 16).max(axis=(2, 4))),
                      ("64 cols x 8 rows", Wb.reshape(n // 8, sx // 64, 64).max(axis=2))):
        tot[name] = tot.get(name, 0.0) + arr.sum()
    B4 = Wb4.reshape(n // 32, 8, sx).reshape(n // 64, 2, 8, sx // 32, 32).transpose(0, 3, 2, 1, 4).reshape(n // 64, sx // 32, 8, 64)
    tot["4-row blocks"] += B4.max(-1).sum() / 2           # (half the rows per block: half the work per step)
base = tot["kernel order"]
print(f"{cfg}: wave steps per block (8 rows), {nsl} slices")
for k, v in tot.items():
    print(f"  {k:18s} {v / (nsl * 512 * 512 / 8 / 64):7.2f}   {v / base:5.2f} x")
