"""Seeded synthetic label volumes for tests, golden fixtures and bench.py.

All generators return arrays whose shape is (sx, sy, sz) in Fortran order when asked for
`order="F"`, i.e. x is the fastest axis, matching the reference's native layout
(reference: src/edt.hpp:434, src/edt.pyx:659-664).  Only numpy is used so that the same
inputs can be rebuilt on the GPU box.
"""
from __future__ import annotations

import numpy as np


def blocky_labels(shape, nlabels=5, zero_frac=0.2, block=4, rng=None):
    """Piecewise-constant random labels in 1..nlabels with a fraction of background blocks."""
    rng = np.random.default_rng(0) if rng is None else rng
    small = tuple(max(1, -(-s // block)) for s in shape)
    a = rng.integers(1, nlabels + 1, size=small)
    a[rng.random(small) < zero_frac] = 0
    for ax in range(len(shape)):
        a = a.repeat(block, axis=ax)
    return np.ascontiguousarray(a[tuple(slice(0, s) for s in shape)])


def blob_mask(shape, rng=None, p=0.6, block=4):
    """Binary blobs: a coarse random field up-sampled by `block`."""
    rng = np.random.default_rng(0) if rng is None else rng
    small = tuple(max(1, -(-s // block)) for s in shape)
    a = (rng.random(small) < p).astype(np.uint8)
    for ax in range(len(shape)):
        a = a.repeat(block, axis=ax)
    return np.ascontiguousarray(a[tuple(slice(0, s) for s in shape)])


def voronoi_labels(shape, nseeds, seed=0, upsample=4, membrane=0.0, dtype=np.uint32):
    """SNEMI3D-like dense segmentation: nearest-seed labels on a coarse grid, up-sampled.

    `membrane` > 0 zeroes that fraction of coarse cells to create thin background sheets
    between segments.
    """
    rng = np.random.default_rng(seed)
    small = tuple(max(1, -(-s // upsample)) for s in shape)
    pts = rng.random((nseeds, len(shape))) * np.array(small)
    grids = np.meshgrid(*[np.arange(s) + 0.5 for s in small], indexing="ij")
    g = np.stack(grids, -1).reshape(-1, len(shape))
    try:  # same image here and on the GPU box; brute force only as a fallback
        from scipy.spatial import cKDTree
        idx = cKDTree(pts).query(g)[1]
    except ImportError:  # pragma: no cover
        idx = np.empty(len(g), dtype=np.int64)
        pts32 = pts.astype(np.float32)
        chunk = max(1, (1 << 24) // max(1, nseeds))
        for s in range(0, len(g), chunk):
            d = ((g[s:s + chunk, None, :].astype(np.float32) - pts32[None, :, :]) ** 2).sum(-1)
            idx[s:s + chunk] = d.argmin(1)
    lab = (idx + 1).reshape(small)
    if membrane > 0:
        lab[rng.random(small) < membrane] = 0
    for ax in range(len(shape)):
        lab = lab.repeat(upsample, axis=ax)
    lab = lab[tuple(slice(0, s) for s in shape)]
    return np.asfortranarray(lab.astype(dtype))


def voronoi_coarse(coarse_shape, nseeds, seed=0, zrange=None, workers=-1):
    """Nearest-seed labels (1..nseeds, uint32, Fortran order) on a coarse grid, optionally only the
    z-range [z0, z1) of it -- the ranks of a Z-sharded run build just their own slab of ONE global
    segmentation (BASELINE configs[3]: 16 000 seeds on 256^3, up-sampled x4 afterwards)."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(seed)
    pts = rng.random((nseeds, 3)) * np.array(coarse_shape)
    z0, z1 = (0, coarse_shape[2]) if zrange is None else zrange
    gx, gy, gz = np.meshgrid(np.arange(coarse_shape[0]) + 0.5, np.arange(coarse_shape[1]) + 0.5,
                             np.arange(z0, z1) + 0.5, indexing="ij")
    g = np.stack([gx, gy, gz], -1).reshape(-1, 3)
    idx = cKDTree(pts).query(g, workers=workers)[1]
    return np.asfortranarray((idx + 1).astype(np.uint32).reshape(coarse_shape[0], coarse_shape[1], z1 - z0))


def voronoi_full(shape, nseeds, seed=0, dtype=np.uint32, workers=-1):
    """FULL-resolution nearest-seed segmentation (smooth cell walls, no up-sampling): large label runs along every
    axis -- ~60 seeds in 512^3 give cells ~130 voxels across, ~500 seeds ~65.  Built slab by slab (bounded memory)."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(seed)
    pts = rng.random((nseeds, 3)) * np.array(shape)
    tree = cKDTree(pts)
    out = np.empty(shape, dtype=dtype, order="F")
    gx, gy = np.meshgrid(np.arange(shape[0]) + 0.5, np.arange(shape[1]) + 0.5, indexing="ij")
    zstep = max(1, (1 << 22) // max(1, shape[0] * shape[1]))
    for z0 in range(0, shape[2], zstep):
        z1 = min(shape[2], z0 + zstep)
        g = np.empty((shape[0], shape[1], z1 - z0, 3))
        g[..., 0] = gx[:, :, None]
        g[..., 1] = gy[:, :, None]
        g[..., 2] = (np.arange(z0, z1) + 0.5)[None, None, :]
        idx = tree.query(g.reshape(-1, 3), workers=workers)[1]
        out[:, :, z0:z1] = (idx + 1).reshape(shape[0], shape[1], z1 - z0)
    return out


# large-cell segmentations (the regime where the hull path / long windows were slow)
LARGE_CELL = {"cfg3L": (60, (1.0, 1.0, 1.0)), "cfg3La": (60, (6.0, 6.0, 30.0)),
              "cfg3M": (500, (1.0, 1.0, 1.0)), "cfg3Ma": (500, (6.0, 6.0, 30.0))}


def sphere_labels(shape, radius, dtype=np.uint32):
    """ONE object: a ball of the given radius (label 1) in the middle of a background volume."""
    ax = [np.arange(s, dtype=np.float32) - (s - 1) / 2.0 for s in shape]
    r2 = (ax[0] ** 2)[:, None, None] + (ax[1] ** 2)[None, :, None] + (ax[2] ** 2)[None, None, :]
    return np.asfortranarray((r2 <= np.float32(radius) ** 2).astype(dtype))


def diagonal_halves(shape, dtype=np.uint32):
    """Two labels separated by the plane x + y + z = const through the middle of the volume: the row values of every pass
    change from row to row (nothing is flat) and the distances are as large as the volume allows."""
    i = [np.arange(s, dtype=np.int32) for s in shape]
    t = i[0][:, None, None] + i[1][None, :, None] + i[2][None, None, :]
    return np.asfortranarray((1 + (t >= (sum(shape) - 3) // 2)).astype(dtype))


# The OBJECT-SIZE sweep: the cost of the column passes as the objects grow -- Voronoi cells ~26 ... ~256
# voxels across, one ball of radius 250, a box without any boundary, a box with ONE background voxel (every z-column sees one
# finite row), two half spaces cut diagonally -- name: (what, anisotropy, black_border)
SWEEP = {
    "sw26": ("voronoi", 7600, (1.0, 1.0, 1.0), False), "sw65": ("voronoi", 500, (1.0, 1.0, 1.0), False),
    "sw130": ("voronoi", 60, (1.0, 1.0, 1.0), False), "sw256": ("voronoi", 8, (1.0, 1.0, 1.0), False),
    "sphere250": ("sphere", 250, (1.0, 1.0, 1.0), False), "sphere250bb": ("sphere", 250, (1.0, 1.0, 1.0), True),
    "onesF": ("ones", 0, (1.0, 1.0, 1.0), False), "onebg": ("onebg", 0, (1.0, 1.0, 1.0), False),
    "diag": ("diag", 0, (1.0, 1.0, 1.0), True), "diagF": ("diag", 0, (1.0, 1.0, 1.0), False),
    "sphere_slab": ("sphere_slab", 250, (1.0, 1.0, 1.0), False),
}


def sweep_volume(name: str, n: int = 512):
    kind, par, an, bb = SWEEP[name]
    shape = (n, n, n)
    if kind == "voronoi":
        return voronoi_full(shape, max(2, int(round(par * (n / 512.0) ** 3))), seed=3), an, bb
    if kind == "sphere":
        return sphere_labels(shape, par * n / 512.0), an, bb
    if kind == "sphere_slab":  # the 8-GPU slab shape of configs[3]: (2n, 2n, n / 4)
        return sphere_labels((2 * n, 2 * n, n // 4), par * n / 512.0), an, bb
    if kind == "ones":
        return np.ones(shape, dtype=np.uint32, order="F"), an, bb
    if kind == "onebg":
        lab = np.ones(shape, dtype=np.uint32, order="F")
        lab[n // 2, n // 2, n // 2] = 0
        return lab, an, bb
    if kind == "diag":
        return diagonal_halves(shape), an, bb
    raise KeyError(name)


def several_slabs_volume(shape):
    """Multi-label volume of more than 2^27 voxels for the slab-wise passes X / Y (tests/test_gpu_fullsize.py): blocks of a
    seeded size, 5 % background, one whole slice of a single label (rows without a boundary under black_border=False)."""
    rng = np.random.default_rng(sum(shape))
    lab = np.asfortranarray(blocky_labels(shape, nlabels=40, zero_frac=0.05, block=int(rng.integers(16, 70)), rng=rng).astype(np.uint32))
    lab[:, :, shape[2] // 2] = 77
    return lab


def config_volume(name: str, n: int = 512):
    """The BASELINE.json configurations at edge length `n` (Fortran order, x fastest).

    Returns (labels, anisotropy, black_border).
      cfg1: all-ones uint32, (1,1,1), black_border=True
      cfg2: all-ones uint32 single label, (6,6,30), black_border=True      <- headline metric
      cfg3: ~2000 (scaled with volume) random multi-labels, black_border=False
      cfg3f: the same labels at anisotropy (3.58, 3.58, 40)
      cfg4: the same kind of segmentation as configs[3] builds it (16 000 seeds at 1024^3), black_border=False
      cfg5: uint8 binary blobs, black_border=True
      cfg3L / cfg3La / cfg3M / cfg3Ma: full-resolution Voronoi segmentations with LARGE cells (~130 / ~65 voxels
            across at 512^3) at (1,1,1) / (6,6,30), black_border=False
    """
    if name in SWEEP:
        return sweep_volume(name, n)
    if name == "cfg1":
        return np.ones((n, n, n), dtype=np.uint32, order="F"), (1.0, 1.0, 1.0), True
    if name == "cfg2":
        return np.ones((n, n, n), dtype=np.uint32, order="F"), (6.0, 6.0, 30.0), True
    if name == "cfg3":
        nseeds = max(8, int(round(2000 * (n / 512.0) ** 3)))
        return voronoi_labels((n, n, n), nseeds, seed=0, upsample=4), (1.0, 1.0, 1.0), False
    if name == "cfg3f":  # same labels, voxel sizes whose multiples are not exact in fp32 (a typical EM resolution in nm)
        nseeds = max(8, int(round(2000 * (n / 512.0) ** 3)))
        return voronoi_labels((n, n, n), nseeds, seed=0, upsample=4), (3.58, 3.58, 40.0), False
    if name == "cfg3m":  # same with thin zero membranes
        nseeds = max(8, int(round(2000 * (n / 512.0) ** 3)))
        return voronoi_labels((n, n, n), nseeds, seed=0, upsample=4, membrane=0.05), (6.0, 6.0, 30.0), False
    if name == "cfg4":  # the 1024^3 segmentation of configs[3] (scaled with the volume), seed 1
        nseeds = max(8, int(round(16000 * (n / 1024.0) ** 3)))
        c = max(1, n // 4)
        lab = voronoi_coarse((c, c, c), nseeds, seed=1)
        for ax in range(3):
            lab = lab.repeat(4, axis=ax)
        return np.asfortranarray(lab[:n, :n, :n]), (1.0, 1.0, 1.0), False
    if name in LARGE_CELL:  # full-resolution Voronoi cells: ~130 voxels (cfg3L*) / ~65 voxels (cfg3M*) across at 512^3
        seeds, an = LARGE_CELL[name]
        nseeds = max(4, int(round(seeds * (n / 512.0) ** 3)))
        return voronoi_full((n, n, n), nseeds, seed=3), an, False
    if name == "cfg5":
        rng = np.random.default_rng(5)
        up = 16 if n >= 64 else 4
        m = blob_mask((n, n, n), rng=rng, p=0.6, block=up)
        return np.asfortranarray(m.astype(np.uint8)), (1.0, 1.0, 1.0), True
    raise KeyError(name)


def box_edtsq_closed_form(shape, anisotropy):
    """Exact squared EDT of an all-foreground box with black border (integer anisotropy):
    the nearest background voxel lies straight across the nearest face."""
    out = None
    for ax, (s, w) in enumerate(zip(shape, anisotropy)):
        i = np.arange(s, dtype=np.int64)
        d = np.minimum(i + 1, s - i).astype(np.float64) * float(w)
        d2 = (d * d).astype(np.float32)
        shp = [1] * len(shape)
        shp[ax] = s
        d2 = d2.reshape(shp)
        out = d2 if out is None else np.minimum(out, d2)
    return np.asfortranarray(np.broadcast_to(out, shape))
