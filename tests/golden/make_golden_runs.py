"""Generate tests/golden/edt_runs.npz from the REAL reference Python module (run utilities + each()).

Build-container only (needs /root/reference and `make -C oracle pyref`):

    python tests/golden/make_golden_runs.py

Records, for small seeded label arrays: runs(labels) (flattened to keys / counts / start-end pairs),
the images each() yields for every label (given a float32 image standing in for the DT), and the
results of draw / transfer / erase along one label's runs.
"""
import importlib.util
import os
import sys
import sysconfig

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from synth import blocky_labels  # noqa: E402

EXT = sysconfig.get_config_var("EXT_SUFFIX")
spec = importlib.util.spec_from_file_location("edt", os.path.join(ROOT, "oracle", "_ref", "edt" + EXT))
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def main():
    rng = np.random.default_rng(20240926)
    dtypes = [np.uint8, np.uint16, np.uint32, np.uint64, np.int32, np.int64]
    blob = {}
    ncases = 24
    for t in range(ncases):
        dims = 1 + t % 3
        shape = tuple(int(rng.integers(1, 14)) for _ in range(dims))
        lab = blocky_labels(shape, nlabels=int(rng.integers(1, 6)), zero_frac=float(rng.random() * 0.4),
                            block=int(rng.integers(1, 4)), rng=rng).astype(dtypes[t % len(dtypes)])
        lab = np.asfortranarray(lab) if t % 2 else np.ascontiguousarray(lab)
        dt = (rng.random(shape) * 100).astype(np.float32)
        dt = np.asfortranarray(dt) if t % 2 else np.ascontiguousarray(dt)
        rr = ref.runs(lab)
        keys = np.array(sorted(rr.keys()), dtype=np.int64)
        counts = np.array([len(rr[k]) for k in keys], dtype=np.int64)
        pairs = np.array([p for k in keys for p in rr[k]], dtype=np.int64).reshape(-1, 2)
        pre = f"{t:03d}/"
        blob[pre + "labels"] = lab
        blob[pre + "dt"] = dt
        blob[pre + "keys"] = keys
        blob[pre + "counts"] = counts
        blob[pre + "pairs"] = pairs
        for in_place in (False, True):
            imgs = [(k, np.array(img)) for k, img in ref.each(lab, dt, in_place=in_place)]
            blob[pre + f"each{int(in_place)}_keys"] = np.array([k for k, _ in imgs], dtype=np.int64)
            blob[pre + f"each{int(in_place)}_imgs"] = (np.stack([i for _, i in imgs]) if imgs
                                                       else np.zeros((0,) + shape, np.float32))
        k0 = int(keys[len(keys) // 2])
        canvas = np.zeros_like(lab)
        blob[pre + "draw_key"] = np.array(k0)
        blob[pre + "draw"] = np.array(ref.draw(7, rr[k0], canvas))
        dest = np.full(shape, -1.0, dtype=np.float32, order="F" if t % 2 else "C")
        blob[pre + "transfer"] = np.array(ref.transfer(rr[k0], dt, dest))
        img = dt.copy(order="K")
        blob[pre + "erase"] = np.array(ref.erase(rr[k0], img))
    blob["ncases"] = np.array(ncases)
    path = os.path.join(HERE, "edt_runs.npz")
    np.savez_compressed(path, **blob)
    print(path, ncases, "cases", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
