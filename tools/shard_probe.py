"""1-rank probe of the Z-sharded driver (RCCL, world 1): host enqueue time vs GPU time per step for
different chunk counts -- tells whether chunking costs launches (host) or tails (GPU).
Run: MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 RANK=0 WORLD_SIZE=1 python tools/shard_probe.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "euclidean-distance-transform-3d_amd"))
from edt import _lib, device  # noqa: E402
from edt import distributed as edist  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
ext = tuple(int(v) for v in os.environ.get("PROBE_EXT", "512,512,512").split(","))
labels = torch.ones((ext[2], ext[1], ext[0]), dtype=torch.int32, device=dev)
for chunks in (1, 2, 4, 8):
    plan = edist.ShardedEDT(ext, _lib.U32, chunks=chunks)
    for _ in range(3):
        plan.run(labels, (6.0, 6.0, 30.0), black_border=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        plan.run(labels, (6.0, 6.0, 30.0), black_border=True)
    e1.record()
    host = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    print(f"ext {ext} chunks {chunks}: host enqueue {host:.3f} ms/step, gpu {e0.elapsed_time(e1) / n:.3f} ms/step, wall {wall:.3f}")
    # per-pass GPU times of one step
    device.set_profiling(True)
    lib = _lib.load()
    lib.edt_hip_set_debug_mode(0x1000) if False else None
    device.set_profiling(False)
dist.destroy_process_group()
