#!/usr/bin/env python3
"""How often does a LANE take each slow path of the column kernel, and how often does a WAVE pay for it?
Runs the host lane emulation (tests/lane_stats.cpp) on z-slices of a synthetic segmentation.
usage: python tools/lane_stats.py [cfg3|smooth|ones] [n]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
from oracle import harness
from synth import voronoi_labels
kind = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
csrc = os.path.join(ROOT, "euclidean-distance-transform-3d_amd", "csrc")
so = os.path.join(ROOT, "tests", "_build", "liblane_stats.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-shared", "-fPIC", f"-I{csrc}", f"-I{ROOT}/tests",
                os.path.join(ROOT, "tests", "lane_stats.cpp"), "-o", so], check=True)
lib = ctypes.CDLL(so)
if not harness.have_port():
    harness.build("port")
o = harness.port()
if kind == "ones":
    vol = np.ones((n, n, 8), dtype=np.uint32, order="F")
elif kind == "smooth":
    vol = voronoi_labels((n, n, 8), max(8, 2000 * n * n * 8 // 512 ** 3 * 8), seed=3, upsample=1)
else:
    vol = voronoi_labels((n, n, 8), max(8, 2000 * n * n * 8 // 512 ** 3), seed=0, upsample=4)
names = ["pop", "bridge_call", "bridge_step", "own_row", "general_row", "resync", "prologue_step", "advance",
         "find_next", "find_next_lds", "fresh"]
lib.lane_stats_reset()
wx = wy = 1.0
bb = False
nslices = 0
for z in range(0, vol.shape[2], 2):
    lab = np.ascontiguousarray(vol[:, :, z].T).astype(np.uint32)  # [y][x]
    f = np.stack([o.raw1d(np.ascontiguousarray(r), 2, r.size, wx, bb) for r in lab]).astype(np.float32)
    f[np.isinf(f)] = np.float32(3.402823466e+38)
    rc = lib.lane_emul_column_pass(lab.ctypes.data_as(ctypes.c_void_p), f.ctypes.data_as(ctypes.c_void_p),
                                   ctypes.c_int64(lab.shape[1]), ctypes.c_int64(lab.shape[0]), ctypes.c_float(wy),
                                   ctypes.c_int(int(bb)), ctypes.c_int(1))
    assert rc == 0
    nslices += 1
buf = (ctypes.c_double * (2 * len(names)))()
lib.lane_stats_get(buf, len(names))
nb = (lab.shape[0] + 31) // 32
cw = 32 if nb <= 2 else 16 if nb <= 4 else 8 if nb <= 8 else 4 if nb <= 16 else 2
lib.lane_stats_grouping(1, cw)
buf2 = (ctypes.c_double * (2 * len(names)))()
lib.lane_stats_get(buf2, len(names))
lanes_total = nslices * lab.shape[1] * ((lab.shape[0] + 31) // 32)       # (column, band) pairs
waves_total = lanes_total / 64.0
print(f"{kind} n={n}: {nslices} slices, {lanes_total} lanes, {waves_total:.0f} waves; per 32-row band:")
print(f"{'event':16s} {'per lane':>10s} {'per wave':>10s} {'row-layout':>11s}   (a wave executes a path once per row if ANY lane takes it; loops: max trip count;")
print(f"{'':16s} {'':>10s} {'':>10s} {'':>11s}    row-layout = waves of 32 adjacent columns x 2 bands)")
for i, nm in enumerate(names):
    print(f"{nm:16s} {buf[2*i]/lanes_total:10.2f} {buf[2*i+1]/waves_total:10.2f} {buf2[2*i+1]/waves_total:11.2f}")
