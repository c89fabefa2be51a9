#!/usr/bin/env python3
"""cProfile of the HOST side of the sharded driver (edt/distributed.py) as a 1-rank RCCL job on one GPU: where the time inside
plan.run() goes per step (R16=0: fp32 records -- pure enqueue, no wait for the 16-bit records' agreement).  Round 4, 512^3,
4 chunks: ~0.4 ms of host time per step, of which 4 x ~65 us are dist.all_to_all and 4 x ~37 us the XY phase's launches."""
import os, sys, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "euclidean-distance-transform-3d_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import numpy as np, torch, torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1)
import bench
from edt import _lib
from edt.distributed import ShardedEDT
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
ext = (512, 512, 512)
plan = ShardedEDT(ext, _lib.U32, reuse_output=True, chunks=4, records16=(os.environ.get("R16", "1") == "1"))
labels = bench.slab_labels(ext, 0, 512, dev, "cfg4")
for _ in range(5): plan.run(labels, (1.0, 1.0, 1.0), black_border=False)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): plan.run(labels, (1.0, 1.0, 1.0), black_border=False)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
dist.destroy_process_group()
