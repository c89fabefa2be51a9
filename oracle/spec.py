"""oracle/spec.py -- TEST INFRASTRUCTURE ONLY.

Algorithm-independent statement of the multi-label anisotropic squared EDT as a brute-force
numpy program (tiny inputs only).  It defines WHAT every implementation -- the reference, the
C restatement in edt_oracle.c and the HIP kernels -- must produce:

  pass 1 (x): for each maximal run [s,e] of one non-zero label in a row,
        d = min(T[i-s+1] if (s>0 or bb) else inf,  T[e-i+1] if (e<n-1 or bb) else inf),
        T[0]=0, T[k]=fl32(T[k-1]+w);  F = fl32(d*d)            (reference: src/edt.hpp:70-119)
  !bb: F=inf -> FLT_MAX                                          (src/edt.hpp:39-45)
  pass 2,3: for each maximal run [a,b] of one non-zero label in a column,
        m = fl32(min_j w2*(p-j)^2 + F[j]) in fp64, w2 = fp64(fl32(w*w)); then min with the
        border parabolas where a border exists                    (src/edt.hpp:168-377)
  !bb: F>=FLT_MAX -> inf                                         (src/edt.hpp:47-53)
"""
import numpy as np

FMAX = np.float32(np.finfo(np.float32).max)


def _runs(col):
    n = len(col)
    s = 0
    for i in range(1, n + 1):
        if i == n or col[i] != col[s]:
            yield s, i - 1, col[s]
            s = i


def _x_row(seg, w, bb):
    n = len(seg)
    w = np.float32(w)
    T = np.zeros(n + 2, dtype=np.float32)
    for k in range(1, n + 2):
        T[k] = np.float32(T[k - 1] + w)
    out = np.zeros(n, dtype=np.float32)
    # pass-1 runs: background separates, and equal labels across a background gap are
    # distinct runs as well
    for s, e, lab in _runs(seg):
        if lab == 0:
            continue
        for i in range(s, e + 1):
            left = T[i - s + 1] if (s > 0 or bb) else np.float32(np.inf)
            right = T[e - i + 1] if (e < n - 1 or bb) else np.float32(np.inf)
            d = np.float32(min(left, right))
            with np.errstate(over="ignore"):
                out[i] = np.float32(d * d)
    return out


def _column(seg, f, w, bb):
    n = len(seg)
    w2 = np.float64(np.float32(w) * np.float32(w))
    out = f.copy()
    for a, b, lab in _runs(seg):
        if lab == 0:
            continue
        j = np.arange(a, b + 1, dtype=np.float64)
        fj = f[a:b + 1].astype(np.float64)
        for p in range(a, b + 1):
            m = np.float32(np.min(w2 * (p - j) ** 2 + fj))
            if bb or a > 0:
                m = min(m, np.float32(w2 * np.float64(p - a + 1) ** 2))
            if bb or b < n - 1:
                m = min(m, np.float32(w2 * np.float64(b - p + 1) ** 2))
            out[p] = m
    return out


def edtsq_xfast(labels, weights, bb):
    """labels: ndarray indexed [x, y, z] (any memory order); returns float32 [x, y, z]."""
    lab = np.asarray(labels)
    nd = lab.ndim
    lab3 = lab.reshape(lab.shape + (1,) * (3 - nd))
    sx, sy, sz = lab3.shape
    F = np.zeros(lab3.shape, dtype=np.float32)
    for z in range(sz):
        for y in range(sy):
            F[:, y, z] = _x_row(lab3[:, y, z], weights[0], bb)
    if nd >= 2:
        if not bb:
            F[np.isinf(F)] = FMAX
        for z in range(sz):
            for x in range(sx):
                F[x, :, z] = _column(lab3[x, :, z], F[x, :, z], weights[1], bb)
        if nd >= 3:
            for y in range(sy):
                for x in range(sx):
                    F[x, y, :] = _column(lab3[x, y, :], F[x, y, :], weights[2], bb)
        if not bb:
            F[F >= FMAX] = np.inf
    return F.reshape(lab.shape)
