"""Generate tests/golden/*.npz from the REAL reference Python module.

Run in the build container only (needs /root/reference and `make -C oracle pyref`):

    python tests/golden/make_golden.py

It imports the reference's Cython module (compiled from /root/reference/src/edt.pyx with the
reference's own flags into oracle/_ref/) and records inputs + outputs of its public API for
a set of small seeded cases.  The .npz files are committed; the GPU box and the CPU test
suite only read them (they never import the reference).
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import sysconfig  # noqa: E402

from synth import blocky_labels, blob_mask  # noqa: E402

EXT = sysconfig.get_config_var("EXT_SUFFIX")
spec = importlib.util.spec_from_file_location("edt", os.path.join(ROOT, "oracle", "_ref", "edt" + EXT))
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def pack(cases, path):
    blob = {}
    for i, c in enumerate(cases):
        for k, v in c.items():
            blob[f"{i:03d}/{k}"] = np.asarray(v)
    np.savez_compressed(path, **blob)
    print(path, len(cases), "cases", os.path.getsize(path) // 1024, "KiB")


def main():
    rng = np.random.default_rng(20240925)
    dtypes = [np.uint8, np.uint16, np.uint32, np.uint64, np.int32, np.float32, np.float64, bool]
    anisos = [(1, 1, 1), (6, 6, 30), (4, 4, 40), (0.5, 0.7, 1.3), (3, 1, 2)]

    # ---- edtsq / edt on random multi-label arrays, all dims / orders / dtypes ---------
    cases = []
    for t in range(96):
        dims = int(rng.integers(1, 4))
        shape = tuple(int(rng.integers(1, 34)) for _ in range(dims))
        dtype = dtypes[t % len(dtypes)]
        lab = blocky_labels(shape, nlabels=int(rng.integers(1, 7)), zero_frac=float(rng.random() * 0.4),
                            block=int(rng.integers(1, 6)), rng=rng).astype(dtype)
        order = "F" if rng.random() < 0.5 else "C"
        lab = np.asfortranarray(lab) if order == "F" else np.ascontiguousarray(lab)
        an = anisos[int(rng.integers(0, len(anisos)))][:dims]
        bb = bool(rng.integers(0, 2))
        an_arg = an[0] if dims == 1 else an
        cases.append(dict(labels=lab, order=order, anisotropy=np.array(an, dtype=np.float64),
                          black_border=bb,
                          edtsq=ref.edtsq(lab, anisotropy=an_arg, black_border=bb),
                          edt=ref.edt(lab, anisotropy=an_arg, black_border=bb)))
    pack(cases, os.path.join(HERE, "edt_random.npz"))

    # ---- one mid-size volume per BASELINE config family (F order, x fastest) ----------
    cases = []
    lab = np.ones((48, 40, 36), dtype=np.uint32, order="F")
    for an, bb in (((1, 1, 1), True), ((6, 6, 30), True), ((6, 6, 30), False)):
        cases.append(dict(labels=lab, order="F", anisotropy=np.array(an, float), black_border=bb,
                          edtsq=ref.edtsq(lab, anisotropy=an, black_border=bb)))
    lab = np.asfortranarray(blocky_labels((56, 48, 40), nlabels=40, zero_frac=0.05, block=6, rng=rng)
                            .astype(np.uint32))
    for an, bb in (((1, 1, 1), False), ((6, 6, 30), False), ((4, 4, 40), True)):
        cases.append(dict(labels=lab, order="F", anisotropy=np.array(an, float), black_border=bb,
                          edtsq=ref.edtsq(lab, anisotropy=an, black_border=bb)))
    pack(cases, os.path.join(HERE, "edt_configs.npz"))

    # ---- sdf and voxel_graph ------------------------------------------------------------
    cases = []
    for t in range(10):
        dims = 2 + (t % 2)
        shape = tuple(int(rng.integers(3, 24)) for _ in range(dims))
        m = blob_mask(shape, rng=rng, p=0.55, block=3).astype([np.uint8, np.uint32, np.uint16][t % 3])
        if t % 2:
            m = np.asfortranarray(m)
        an = anisos[t % len(anisos)][:dims]
        bb = bool(t % 3 == 0)
        cases.append(dict(kind="sdf", labels=m, anisotropy=np.array(an, float), black_border=bb,
                          out=ref.sdf(m, anisotropy=an, black_border=bb)))
    for t in range(14):
        dims = 2 + (t % 2)
        shape = tuple(int(rng.integers(2, 16)) for _ in range(dims))
        m = blob_mask(shape, rng=rng, p=0.7, block=2).astype([np.uint8, np.uint32, bool][t % 3])
        g = rng.integers(0, 64, size=shape).astype(np.uint8)
        g[rng.random(shape) < 0.6] = 0b00111111
        if t % 2:
            m, g = np.asfortranarray(m), np.asfortranarray(g)
        an = anisos[t % len(anisos)][:dims]
        bb = bool(t % 2 == 0)
        cases.append(dict(kind="voxel_graph", labels=m, graph=g, anisotropy=np.array(an, float),
                          black_border=bb,
                          out=ref.edtsq(m, anisotropy=an, black_border=bb, voxel_graph=g)))
    pack(cases, os.path.join(HERE, "edt_sdf_voxel_graph.npz"))


if __name__ == "__main__":
    main()
