#!/usr/bin/env python3
"""The OBJECT-SIZE sweep: what the column passes cost as the objects grow.  For every volume of
tests/synth.py: SWEEP (Voronoi cells ~26 ... ~256 voxels across, one ball of radius 250, a box without any boundary, a box
with ONE background voxel, two half spaces cut diagonally, the ball on the 8-GPU slab shape) one line: ms per step, the passes,
the 32 B/voxel model fraction, and `verified` = the timed output bit for bit against the compiled reference (all threads).

usage: python tools/object_sweep.py [names,comma,separated] [modes,comma,separated]     (modes: EDT_HIP_DEBUG_MODE values,
       default 0; e.g. 0,0x20000000 = also without the wide form: fp32 hand-over, 0x8000000 = fp32 kernels only)
Writes gpurun_out/object_sweep.json (a list of the lines)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "euclidean-distance-transform-3d_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from edt import _lib  # noqa: E402
from synth import SWEEP  # noqa: E402


def main():
    names = [a for a in (sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] else sorted(SWEEP))]
    modes = [int(m, 0) for m in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0"])]
    n = int(os.environ.get("SWEEP_SIZE", "512"))
    steps = int(os.environ.get("SWEEP_STEPS", "30"))
    verify = os.environ.get("SWEEP_VERIFY", "1") != "0"
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ref = kind = None
    if verify:
        ref, kind = bench.reference_lib()
    lines = []
    for name in names:
        run = bench.DeviceRun(name, n, dev)
        want = None
        for mode in modes:
            lib.edt_hip_set_debug_mode(mode)
            s, kern, _ = run.measure(steps, 3)
            line = {"config": name, "shape": list(run.shape), "anisotropy": list(run.an), "black_border": run.bb, "mode": hex(mode), **s}
            if verify:
                if want is None:
                    threads = (os.cpu_count() or 1,) if kind == "reference" else (1,)
                    res, want = bench.time_reference(ref, kind, run.host_labels(), tuple(run.an), run.bb, threads)
                    line["reference_mvox_per_s"] = {str(k): round(v, 1) for k, v in res.items()}
                line["verified"] = bool(np.array_equal(run.out.cpu().numpy().reshape(-1), np.asarray(want).reshape(-1)))
            lines.append(line)
            print(json.dumps(line), flush=True)
        lib.edt_hip_set_debug_mode(0)
        del run, want
        torch.cuda.empty_cache()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "object_sweep.json"), "w") as f:
        json.dump(lines, f, indent=1)


if __name__ == "__main__":
    main()
