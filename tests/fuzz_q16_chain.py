#!/usr/bin/env python3
"""Randomised CPU parity of passes Y and Z of whole 3-D volumes through the integer kernel's lane logic (tests/q16_emul.cpp, the
header the HIP kernel is built from) against the oracle, with the host's proof "no tile can be refused" (csrc/edt_api.hip:
q16_cannot_refuse, restated in tests/test_q16_logic.py) checked on every case.  No GPU.  Test infrastructure, not collected by
pytest (minutes per seed).  usage: python tests/fuzz_q16_chain.py <seed> <ncases>   (run tests/test_q16_logic.py once before: it
builds tests/_build/libq16_emul.so)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np, ctypes
import test_q16_logic as T
from oracle import harness
from synth import blocky_labels
o = harness.port()
lib = ctypes.CDLL(os.path.join(ROOT, 'tests', '_build', 'libq16_emul.so'))
lib.q16_emul_column_pass.restype = ctypes.c_int
lib.q16_emul_quantum.restype = ctypes.c_int
seed = int(sys.argv[1]); ncase = int(sys.argv[2])
rng = np.random.default_rng(seed)
bad = 0
for ci in range(ncase):
    sx = 4 * int(rng.integers(1, 30)); sy = int(rng.integers(97, 700)); sz = int(rng.integers(97, 700))
    if sx * sy * sz > 6e6: continue
    kind = rng.integers(0, 4)
    if kind == 0: lab = np.ones((sz, sy, sx), np.uint32)
    elif kind == 1:
        lab = np.ones((sz, sy, sx), np.uint32); lab[:, :int(rng.integers(1, 5)), :] = 2; lab[:int(rng.integers(1, 4)), :, :sx // 2] = 3
        for _ in range(int(rng.integers(0, 4))): lab[rng.integers(0, sz), rng.integers(0, sy), rng.integers(0, sx)] = 0
    else:
        lab = blocky_labels((sz, sy, sx), nlabels=int(rng.integers(1, 6)), zero_frac=float(rng.random() * 0.1), block=int(rng.integers(5, 120)), rng=rng).astype(np.uint32)
    w = tuple(float(v) for v in rng.choice([1, 2, 6, 30, 0.5, 4, 40, 3, 10], size=3))
    bb = bool(rng.random() < 0.25)
    ok, q, a = T.quantum(lib, w)
    if not ok: continue
    sure_y, sure_z = T.host_cannot_refuse(a, q, (sx, sy, sz), bb)
    want = o.raw3d(np.ascontiguousarray(lab).reshape(-1), 2, sx, sy, sz, w, bb).reshape(sz, sy, sx)
    after_y = np.empty((sz, sy, sx), np.float32)
    ny = nz = ay = az = 0
    err = None
    for z in range(sz):
        _, codes = T.x_pass(o, lab[z], w[0], bb)
        w2 = o.raw2d(np.ascontiguousarray(lab[z]).reshape(-1), 2, sx, sy, (w[0], w[1]), bb).reshape(sy, sx)
        w2 = np.where(np.isinf(w2), np.float32(T.FLT_MAX), w2).astype(np.float32)
        got, tiles = T.column_pass(lib, lab[z], None, codes, q, a[1], a[0], bb, 0)
        if sure_y and not tiles.all(): err = ("proofY", z)
        for i, t in enumerate(tiles):
            sl = slice(32 * i, min(sx, 32 * i + 32))
            ny += 1; ay += int(t != 0)
            if t and not np.array_equal(got[:, sl], w2[:, sl]): err = ("Y", z, i, int(t))
        after_y[z] = w2
    for y in range(sy):
        got, tiles = T.column_pass(lib, lab[:, y, :], after_y[:, y, :], None, q, a[2], a[0], bb, 0 if bb else 1)
        if sure_z and not tiles.all(): err = ("proofZ", y)
        for i, t in enumerate(tiles):
            sl = slice(32 * i, min(sx, 32 * i + 32))
            nz += 1; az += int(t != 0)
            if t and not np.array_equal(got[:, sl], want[:, y, sl]): err = ("Z", y, i, int(t))
    print(ci, (sx, sy, sz), kind, w, bb, "sure", sure_y, sure_z, "accepted", f"{ay}/{ny} {az}/{nz}", "ERR %s" % (err,) if err else "ok", flush=True)
    bad += err is not None
print("bad", bad)
