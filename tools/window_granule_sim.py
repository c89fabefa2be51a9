#!/usr/bin/env python3
"""Host models (no GPU) of two remedies for the far windows of the integer column kernel (csrc/experiments/q16_granule_r05):

  1. how many one-sided GRANULES (8 rows of a column pair at offset +-k blocks) a block / a wave has to look at when a granule
     is skipped unless c_{8k-7} + (a lower bound of its smallest value) is below some current minimum -- against the
     symmetric steps that run until c_d alone reaches the minima (round 4);
  2. what dealing the blocks of a tile (or of a wave) to the lanes SORTED by a cheap proxy of their window length would save --
     the remedy of VERDICT r4 item 1 that was priced and NOT built: the proxy needs the border stage of every block (~200
     instructions) before the deal and the results leave the lanes scattered.

Y pass of z-slices of a 512^3 configuration, a = 1.  usage: python tools/window_granule_sim.py [cfg3|cfg3M|cfg3L] [slices]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
from synth import config_volume

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3L"
nsl = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a = 1
INF = 1 << 20
BIG = INF * INF


def xpass(L):  # L[x][y] -> N[y][x] = k^2 of pass X (black_border off: no border at the volume's edge)
    sx, sy = L.shape
    N = np.empty((sy, sx), dtype=np.int64)
    idx = np.arange(sx)
    for y in range(sy):
        row = L[:, y]
        ch = np.flatnonzero(row[1:] != row[:-1]) + 1
        starts = np.concatenate(([0], ch)); ends = np.concatenate((ch, [sx]))
        k = np.full(sx, INF, dtype=np.int64)
        for s, e in zip(starts, ends):
            i = idx[s:e]
            dl = (i - s + 1) if s > 0 else np.full(e - s, INF)
            dr = (e - i) if e < sx else np.full(e - s, INF)
            k[s:e] = np.minimum(dl, dr)
            if row[s] == 0:
                k[s:e] = 0
        N[y] = np.where(k >= INF, INF, k * k)
    return N


def borders(N, L):
    sy, sx = N.shape
    Lt = L.T
    rs = np.ones((sy, sx), bool); rs[1:] = Lt[1:] != Lt[:-1]
    ypos = np.arange(sy)[:, None]
    last = np.maximum.accumulate(np.where(rs, ypos, -1), 0)
    nxt = np.minimum.accumulate(np.where(rs, ypos, sy + INF)[::-1], 0)[::-1]
    nxt2 = np.empty_like(nxt); nxt2[:-1] = nxt[1:]; nxt2[-1] = sy + INF
    d = np.minimum(np.where(last > 0, ypos - last + 1, INF), np.where(nxt2 < sy, nxt2 - ypos, INF))
    return np.where(Lt == 0, 0, np.minimum(N, np.where(d >= INF, BIG, a * d * d)))


lab, an, bb = config_volume(cfg, 512)
acc = {}
for z in np.linspace(20, 490, nsl).astype(int):
    L = np.ascontiguousarray(lab[:, :, z]); N = xpass(L); sy, sx = N.shape; B = borders(N, L)
    nblk = sy // 8
    blockmax = lambda v: v.reshape(nblk, 8, sx // 2, 2).max(axis=(1, 3))
    # --- the kernel's exit rule: executed steps per block, per wave (16 pairs x 4 blocks), sorted dealings -----------------
    Dmax = 160
    pad = np.full((Dmax + 16, sx), BIG, dtype=np.int64)
    Np = np.concatenate((pad, N, pad), 0)
    best = B.copy(); bmax = blockmax(best); bmax0 = bmax.copy()
    ex = np.full((nblk, sx // 2), -1, dtype=np.int64)
    for D in range(1, Dmax + 1):
        if D > 1 and (D - 1) % 8 == 0:
            bmax = blockmax(best)
        if D % 2 == 1:
            done = (bmax <= a * D * D) & (ex < 0); ex[done] = D
        best = np.minimum(best, a * D * D + np.minimum(Np[Dmax + 16 - D: Dmax + 16 - D + sy], Np[Dmax + 16 + D: Dmax + 16 + D + sy]))
    ex[ex < 0] = Dmax
    ex -= 1
    waves = lambda v: v.reshape(nblk // 4, 4, sx // 32, 16).transpose(0, 2, 1, 3).reshape(nblk // 4, sx // 32, 64)
    r = {"steps a block needs (mean)": ex.mean(), "steps a wave executes, kernel's order": waves(ex).max(-1).mean()}
    P = np.minimum(np.ceil(np.sqrt(bmax0 / a)), 255)  # the proxy: the window the border stage's bound asks for
    Et = ex.reshape(nblk, sx // 32, 16).transpose(1, 0, 2).reshape(sx // 32, -1)
    Pt = P.reshape(nblk, sx // 32, 16).transpose(1, 0, 2).reshape(sx // 32, -1)
    grouped = lambda E, o: np.take_along_axis(E, o, -1).reshape(E.shape[:-1] + (-1, 64)).max(-1).mean()
    r["... blocks of a TILE dealt sorted by their true need"] = grouped(Et, np.argsort(Et, 1, kind="stable"))
    r["... blocks of a TILE dealt sorted by the proxy"] = grouped(Et, np.argsort(Pt, 1, kind="stable"))
    nband = nblk // 4
    E4, P4 = ex.reshape(nband, 4, sx // 32, 16), P.reshape(nband, 4, sx // 32, 16)
    idx = [[w + 4 * k for k in range(nband // 4)] for w in range(4)]
    Ew = np.stack([E4[i].transpose(2, 0, 1, 3).reshape(sx // 32, -1) for i in idx], 1)
    Pw = np.stack([P4[i].transpose(2, 0, 1, 3).reshape(sx // 32, -1) for i in idx], 1)
    r["... blocks of a WAVE (its 4 bands) dealt sorted by the proxy"] = grouped(Ew, np.argsort(Pw, -1, kind="stable"))
    # --- granules: relevant one-sided granules per block / looked at by a wave ---------------------------------------------
    R = best  # converged minima
    bmaxR = blockmax(R)
    Ng = N.reshape(nblk, 8, sx // 2, 2)
    code = np.floor(np.sqrt(np.minimum(Ng.min(axis=(1, 3)), 65535)))
    gl = code * code  # the lower bound the kernel uses
    for name, bm_ in (("initial minima", bmax0), ("converged minima", bmaxR)):
        lane_std = lane_g = 0.0; wave_std = wave_g = 0
        for k in range(1, 40):
            dd = a * (8 * k - 7) ** 2
            for sgn in (1, -1):
                j = np.arange(nblk) + sgn * k
                ok = (j >= 0) & (j < nblk)
                G = np.where(ok[:, None], gl[np.clip(j, 0, nblk - 1)], BIG)
                s_, g_ = dd < bm_, dd + G < bm_
                lane_std += s_.mean(); lane_g += g_.mean()
                wave_std += waves(s_).any(-1).mean(); wave_g += waves(g_).any(-1).mean()
        r[f"granules per block, c_d alone ({name})"] = lane_std
        r[f"granules per block, with the granule bound ({name})"] = lane_g
        r[f"granules a wave looks at, c_d alone ({name})"] = wave_std
        r[f"granules a wave looks at, with the granule bound ({name})"] = wave_g
    for k_, v in r.items():
        acc.setdefault(k_, []).append(v)
print(f"{cfg}: Y pass, a = 1, {nsl} slices; a granule = 8 rows of a column pair on ONE side (8 symmetric steps = 2 granules)")
for k_, v in acc.items():
    print(f"  {k_:72s} {np.mean(v):7.2f}")
