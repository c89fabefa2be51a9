#!/usr/bin/env python3
"""Randomised GPU-vs-oracle fuzz over shapes / dtypes / label structures (diagnostics; the regular
parity tests live in tests/).  usage: [FUZZ_Q16=1 [FUZZ_INF=1 | FUZZ_FLAT=1] [FUZZ_PAD=1] | FUZZ_VG=1] python tools/fuzz_gpu.py [ncases] [seed]      (FUZZ_DUMP=1: triage a mismatch under the
form-selection bits and save the case to gpurun_out/)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "euclidean-distance-transform-3d_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import edt
from oracle import harness
from synth import blocky_labels
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
if not harness.have_port():
    harness.build("port")
o = harness.port()
dtypes = [np.uint8, np.uint16, np.uint32, np.uint64, np.float32, bool]
bad = 0
t0 = time.time()
for i in range(ncases):
    dims = 3 if rng.random() < 0.8 else 2
    big = int(rng.integers(0, dims))
    shape = []
    for d in range(dims):
        hi = int(os.environ.get("FUZZ_MAX_AXIS", "1100")) if d == big else 90
        s = int(rng.integers(1, hi))
        if rng.random() < 0.5:
            s = max(4, s // 4 * 4)
        shape.append(s)
    shape = tuple(shape)
    q16 = os.environ.get("FUZZ_Q16") == "1"
    if q16:
        # shapes and voxel sizes of the 16-bit integer column kernel (csrc/edt_colq16.hip): rows of whole 16-byte granules,
        # both column axes of at least four bands (some beyond 512 rows), voxel sizes that share a quantum
        dims = 3
        shape = (4 * int(rng.integers(1, 70)), int(rng.integers(97, 700)), int(rng.integers(97, 400)))
        if rng.random() < 0.3:
            shape = (shape[0], shape[2], shape[1])
        if rng.random() < 0.15:
            shape = (4 * int(rng.integers(1, 12)), int(rng.integers(513, 1100)), int(rng.integers(97, 200)))
    vg = os.environ.get("FUZZ_VG") == "1"
    if vg:
        # the voxel-graph transform (csrc/edt_voxel_graph.hip): doubled column axes of 8..1200 rows, rows of whole granules
        # (index form) and not, random cut links
        dims = 3 if rng.random() < 0.8 else 2
        shape = [int(rng.integers(1, 40)) * (4 if rng.random() < 0.6 else 1) + int(rng.integers(0, 2)) * int(rng.random() < 0.3)]
        shape += [int(rng.integers(4, 600 if rng.random() < 0.2 else 120)) for _ in range(dims - 1)]
        shape = tuple(shape)
    finf = os.environ.get("FUZZ_INF") == "1"
    if np.prod(shape) > (3e7 if q16 else 4e5 if vg else 6e6):
        continue
    kind = rng.integers(0, 3)
    if finf:
        # Volumes of +inf (round 6): one object with SPARSE structure and no black border -- most x-rows see no boundary at all,
        # whole tiles are nothing but +inf, every column holds a few finite rows at most; sometimes a large voxel size along x, so
        # that the few finite values leave the integer form and their tiles are refused.  Targets: tiles answered from the fill, the
        # +inf rows pass Y leaves in the 16-bit plane, windows that start / stop at the finite rows, refused tiles with such rows.
        lab = np.ones(shape, dtype=np.uint32)
        for _ in range(int(rng.integers(0, 6))):
            c = [int(rng.integers(0, s)) for s in shape]
            how = rng.integers(0, 5)
            val = 0 if rng.random() < 0.6 else int(rng.integers(2, 5))
            if how == 0:
                lab[c[0], c[1], c[2]] = val                                   # one voxel
            elif how == 1:
                lab[c[0], :, c[2]] = val                                      # a line along y
            elif how == 2:
                lab[c[0], c[1], :] = val                                      # a line along z
            elif how == 3:
                lab[:, :, c[2]:c[2] + int(rng.integers(1, 4))] = val          # whole slices of another label: +inf along x AND y
            else:
                lab[:, c[1]:, :] = val                                        # a half space along y
        kind = -1
    fflat = os.environ.get("FUZZ_FLAT") == "1"
    if fflat and not finf and dims == 3:
        # Volumes of SLABS and BOXES (round 6): columns that are constant along whole axes -- tiles without a run start behind row 0
        # whose rows all equal row 0 are answered from their image (csrc/edt_colq16.hip: "tiles without structure") -- next to
        # tiles that just are not: a slab along one axis (flat along the other two), a box, single voxels, an x-range whose rows
        # differ from their neighbours' but not along y or z.  Both border rules.
        lab = np.ones(shape, dtype=np.uint32)
        for _ in range(int(rng.integers(0, 5))):
            c = [int(rng.integers(0, s)) for s in shape]
            e = [int(rng.integers(1, max(2, s // 2))) for s in shape]
            how = rng.integers(0, 6)
            val = 0 if rng.random() < 0.4 else int(rng.integers(2, 5))
            if how == 0:
                lab[c[0]:c[0] + e[0], :, :] = val                              # an x-range of every row: flat along y and z
            elif how == 1:
                lab[:, c[1]:c[1] + e[1], :] = val                              # a slab along y
            elif how == 2:
                lab[:, :, c[2]:c[2] + e[2]] = val                              # a slab along z
            elif how == 3:
                lab[c[0]:c[0] + e[0], c[1]:c[1] + e[1], c[2]:c[2] + e[2]] = val  # a box
            elif how == 4:
                lab[c[0], c[1], c[2]] = val                                    # one voxel
            else:
                lab[c[0]:, c[1]:, :] = val                                     # a quadrant: flat along z only
        kind = -1
    if os.environ.get("FUZZ_PAD") == "1":
        # the pitch of the index buffer / 16-bit plane (csrc/edt_api.hip: plane_pad_elems), read per plan
        pad = rng.choice(["", "0", "8", "4096", "8200"])
        if pad:
            os.environ["EDT_HIP_PLANE_PAD_BYTES"] = str(pad)
        else:
            os.environ.pop("EDT_HIP_PLANE_PAD_BYTES", None)
    if kind == -1:
        pass
    elif kind == 0:
        lab = np.ones(shape, dtype=np.uint32)
    else:
        lab = blocky_labels(shape, nlabels=int(rng.integers(1, 6)) if not q16 or rng.random() < 0.5 else int(rng.integers(20, 400)),
                            zero_frac=float(rng.random() * 0.3), block=int(rng.integers(1, 50 if not q16 else 120)), rng=rng)
    g = None
    if vg:
        g = np.full(shape, 0b00111111 if dims == 3 else 0b00001111, dtype=np.uint8)
        for bit in (0x01, 0x02, 0x04, 0x08, 0x10, 0x20)[:2 * dims]:
            g[rng.random(shape) < rng.choice([0.0, 0.01, 0.1])] &= np.uint8(~bit & 0xFF)
    dt = dtypes[i % len(dtypes)]
    lab = np.asfortranarray(lab.astype(dt)) if vg or rng.random() < 0.7 else np.ascontiguousarray(lab.astype(dt))
    if vg:
        g = np.asfortranarray(g)
    an = tuple(float(a) for a in rng.choice([1, 2, 6, 30, 0.5, 4, 40, 3] if q16 else [1, 2, 6, 30, 0.5, 1.3, 7.25], size=dims))
    bb = bool(rng.integers(0, 2))
    if finf:
        bb = rng.random() < 0.1
    if vg:
        if rng.random() < 0.7:
            an = tuple(float(a) for a in rng.choice([1, 2, 6, 30, 4], size=dims))  # (sizes that share a quantum: integer kernel)
        want = o.edtsq(lab, an, bb, voxel_graph=g)
        got = edt.edtsq(lab, anisotropy=an, black_border=bb, voxel_graph=g)
    else:
        want = o.edtsq(lab, an, bb)
        got = edt.edtsq(lab, anisotropy=an, black_border=bb)
    if not np.array_equal(got, want, equal_nan=True):
        bad += 1
        print("MISMATCH", shape, dt.__name__, an, bb, int((got != want).sum()))
        if os.environ.get("FUZZ_DUMP") == "1" and not vg:
            # which form selection answers this case correctly, and where the wrong voxels are; the case itself for the CPU tier
            from edt import _lib
            lib = _lib.load()
            for mode in (0x80, 0x400, 0x480, 0x20000000, 0x10000000, 0x8000000):
                lib.edt_hip_set_debug_mode(mode)
                g2 = edt.edtsq(lab, anisotropy=an, black_border=bb)
                print("   mode", hex(mode), "mismatches", int((g2 != want).sum()))
            lib.edt_hip_set_debug_mode(0)
            w = np.argwhere(got != want)
            print("   order", "F" if lab.flags.f_contiguous else "C", "labels", np.unique(lab)[:8], "first wrong", w[:3].tolist(), "last", w[-3:].tolist(),
                  "got", got[tuple(w[0])], "want", want[tuple(w[0])], "box", w.min(0).tolist(), w.max(0).tolist())
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"fuzz_fail_{i}.npz"), lab=lab, an=np.array(an), bb=bb)
print(f"{ncases} cases, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
