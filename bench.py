#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mvox/s of the 3-D multi-label squared EDT (edt3dsq).

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line.
For N > 1 it is launched under torch.distributed.run, one rank per GPU (RCCL).

  * N = 1 workload = BASELINE.json configs[1]: 512^3 uint32 single label, anisotropy
    (6,6,30), black_border=True, labels and output resident in HBM (no PCIe in the timed region).
  * N > 1 workload = ONE global volume with 512^3 voxels PER GPU (N=8 -> 1024^3, BASELINE
    configs[3]), Z-sharded; the X and Y passes are slab-local, ONE all-to-all (RCCL send/recv
    group over xGMI) re-partitions Z-slabs into Y-slabs before the Z pass.  Weak scaling.
  * a "step" = one complete edtsq of the (local part of the) volume.
  * roofline: dominant kernel's ALGORITHMIC bytes (pass X reads labels + writes
    fp32, passes Y/Z read labels + read/write fp32 -> 8 / 12 / 12 B per uint32 voxel) divided
    by its duration measured with hipEvents inside the library on the launch stream.
  * cpu_baseline: the real reference (oracle/_ref, compiled from the reference sources) on
    this host's cores, whole 512^3 workload, 1 thread and all threads.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "euclidean-distance-transform-3d_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)


def algorithmic_bytes_per_voxel(label_bytes, fused):
    """X reads labels + writes fp32, Y and Z read labels + read/write fp32.  On the fused
    path the bit kernel does pass X's label read and the first column kernel does the rest of X and Y."""
    if fused:
        return {"x_bits": label_bytes, "y_pass": 4 + label_bytes + 8, "z_pass": label_bytes + 8}
    return {"x_pass": label_bytes + 4, "y_pass": label_bytes + 8, "z_pass": label_bytes + 8}


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the PMC passes of the round (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950);
    tools/profile_round.sh collects them, profiles/r01_traffic.json holds the per-kernel result."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(kernel)
    except (OSError, ValueError):
        return None


def cpu_baseline(n, anisotropy, bb):
    """Time the reference CPU implementation on this host (bounded: whole 512^3 job, ~10-20 s)."""
    try:
        from oracle import harness
        if harness.have_ref():
            lib, kind = harness.ref(fast=True), "reference"
        else:
            if not harness.have_port():
                harness.build("port")
            lib, kind = harness.port(), "port"
    except Exception as e:  # pragma: no cover
        return {"value": None, "unit": "Mvox/s", "cores": 0, "kind": "unavailable", "sample": str(e)}
    cores = os.cpu_count() or 1
    m = min(n, 512)
    lab = np.ones((m, m, m), dtype=np.uint32, order="F")
    vox = lab.size
    results = {}
    threads = (1, cores) if kind == "reference" else (1,)
    for p in threads:
        best = float("inf")
        for _ in range(2):
            t0 = time.perf_counter()
            lib.raw3d(lab, 2, m, m, m, anisotropy, bb, parallel=p) if kind == "reference" else \
                lib.raw3d(lab, 2, m, m, m, anisotropy, bb)
            best = min(best, time.perf_counter() - t0)
        results[p] = vox / best / 1e6
    top = max(results, key=lambda k: results[k])
    return {
        "value": round(results[top], 2), "unit": "Mvox/s", "cores": int(top), "kind": kind,
        "sample": f"whole {m}^3 uint32 volume, anisotropy {anisotropy}, best of 2; "
                  + ", ".join(f"{p} thread(s): {v:.1f} Mvox/s" for p, v in results.items()),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=512, help="edge length of the per-GPU volume")
    ap.add_argument("--config", default="cfg2", choices=["cfg1", "cfg2", "cfg3", "cfg3m", "cfg5"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--generic", action="store_true", help="force the size-agnostic fallback kernels")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks")

    import edt  # noqa: F401
    from edt import _lib, device
    from synth import config_volume

    _lib.load()
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if os.environ.get("EDT_BENCH_ONE_GPU") == "1":  # dry run of the N > 1 leg: every rank on cuda:0 (with gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    n = args.size
    if world > 1 or os.environ.get("EDT_BENCH_FORCE_SHARDED") == "1":  # (the env var: 1-rank dry run of the N > 1 leg)
        import torch.distributed as dist
        backend = os.environ.get("EDT_BENCH_BACKEND", "nccl")  # "gloo": dry run, transfers staged through the host
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        from edt import distributed as edist
        return edist.bench_main(args, rank, world, dev)

    lab_np, an, bb = config_volume(args.config, n)
    label_bytes = lab_np.dtype.itemsize
    vox = lab_np.size
    # (sx,sy,sz) Fortran array == contiguous tensor of shape (sz,sy,sx): no copy of the bytes
    labels = torch.from_numpy(np.ascontiguousarray(lab_np.T).view(np.int32 if label_bytes == 4 else np.uint8)).to(dev)
    out = torch.empty((n, n, n), dtype=torch.float32, device=dev)
    plan = device.Plan(lab_np.shape, 2 if label_bytes == 4 else 0, dev)

    def step():
        plan.run(labels, an, black_border=bb, sqrt=False, out=out, force_generic=args.generic)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms_per_step = elapsed / args.steps * 1e3
    mvox = vox / (elapsed / args.steps) / 1e6

    # per-kernel durations with hipEvents on the launch stream (separate profiled steps)
    device.set_profiling(True)
    acc = {}
    prof_steps = max(3, min(args.steps, 10))
    for _ in range(prof_steps):
        step()
        torch.cuda.synchronize()
        for name, ms in device.pass_times():
            acc.setdefault(name, []).append(ms)
    device.set_profiling(False)
    kernels = {k: float(np.mean(v)) for k, v in acc.items()}
    bpv = algorithmic_bytes_per_voxel(label_bytes, "x_bits" in kernels)
    dom = max((k for k in kernels if k in bpv), key=lambda k: kernels[k])
    achieved = bpv[dom] * vox / (kernels[dom] * 1e-3) / 1e9
    total_kernel_ms = sum(kernels.values())
    roofline = {
        "bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": measured_traffic(dom),
        "kernel_ms": {k: round(v, 4) for k, v in kernels.items()},
        "whole_job_algorithmic_GBs": round(sum(bpv.values()) * vox / (total_kernel_ms * 1e-3) / 1e9, 1),
        "whole_job_frac": round(sum(bpv.values()) * vox / (total_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
    }

    # sanity: the timed output is the right answer (closed form for the all-ones box)
    if args.config in ("cfg1", "cfg2"):
        from synth import box_edtsq_closed_form
        ok = bool(np.array_equal(out.cpu().numpy().T, box_edtsq_closed_form(lab_np.shape, an)))
    else:
        ok = None

    result = {
        "metric": "Mvox/s edt3dsq 512^3 uint32", "value": round(mvox, 1), "unit": "Mvox/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64 envelope / f32 storage / u32 labels", "data": "synthetic",
        "config": {"workload": f"{args.config}: {n}^3 uint32 labels, anisotropy {tuple(an)}, "
                               f"black_border={bb}, device-resident in/out",
                   "path": "generic" if args.generic else "default", "output_verified": ok},
        "roofline": roofline,
    }
    if not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(n, tuple(an), bb)
    print(json.dumps(result))


if __name__ == "__main__":
    main()
