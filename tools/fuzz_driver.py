#!/usr/bin/env python3
"""Randomised parity of the WHOLE sharded driver (edt/distributed.py: partition, halo, chunked slab-record XY phase on two
streams, exchange, agreement on the 16-bit records, Z phase, gather-back) with the real kernels: W processes share cuda:0 and
talk over gloo (host-staged transfers), every process walks the same random cases and checks its part against the oracle.
usage: python tools/fuzz_driver.py [world] [ncases] [seed]"""
import os, socket, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "euclidean-distance-transform-3d_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, ncases, seed, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from edt import _lib
    from edt import distributed as edist
    from oracle import harness
    from synth import blocky_labels
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    o = harness.port()
    rng = np.random.default_rng(seed)          # (the same stream of cases on every rank)
    bad = n16 = nfall = nrec = 0
    for i in range(ncases):
        sx = 4 * int(rng.integers(2, 60)) if rng.random() < 0.8 else int(rng.integers(8, 200))
        sy = int(rng.integers(32 * world, 400))
        sz = int(rng.integers(world, 300))
        if sx * sy * sz > 8e6:
            continue
        an = tuple(float(a) for a in rng.choice([1, 2, 6, 30, 4, 3, 0.5, 1.3], size=3))
        bb, sqrt, gather_back = bool(rng.integers(0, 2)), rng.random() < 0.3, rng.random() < 0.3
        chunks = int(rng.integers(1, 5))
        kind = rng.integers(0, 4)
        if kind == 0:
            vol = np.ones((sx, sy, sz), dtype=np.uint32)
        else:
            vol = blocky_labels((sx, sy, sz), nlabels=int(rng.integers(1, 60)), zero_frac=float(rng.random() * 0.2),
                                block=int(rng.integers(2, 90)), rng=rng).astype(np.uint32)
        vol = np.asfortranarray(vol)
        zyx = np.ascontiguousarray(vol.T)
        plan = edist.ShardedEDT((sx, sy, sz), _lib.U32, chunks=chunks, reuse_output=bool(rng.integers(0, 2)))
        zs, ze = plan.local_z()
        slab = torch.from_numpy(zyx[zs:ze].copy().view(np.int32)).to(dev)
        want = o.edtsq(vol, an, bb)
        if sqrt:
            want = np.sqrt(want)
        want = np.ascontiguousarray(want.T)
        ok = True
        for rep in range(2):   # (the second run reuses buffers and, after a fall-back, stays on fp32 records)
            out = plan.run(slab, an, black_border=bb, sqrt=sqrt, gather_back=gather_back).cpu().numpy()
            if gather_back:
                ok = ok and np.array_equal(out, want[zs:ze], equal_nan=True)
            else:
                ys, ye = plan.local_y()
                ok = ok and np.array_equal(out, want[:, ys:ye, :], equal_nan=True)
        nrec += plan.records
        n16 += plan.last_records16
        nfall += plan.fallbacks16
        if not ok:
            bad += 1
            print(f"rank {rank} MISMATCH", (sx, sy, sz), an, bb, sqrt, gather_back, chunks, plan.records, plan.last_records16, flush=True)
    q.put((rank, bad, nrec, n16, nfall))
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    from oracle import harness
    if not harness.have_port():
        harness.build("port")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    t0 = time.time()
    procs = [ctx.Process(target=worker, args=(r, world, port, ncases, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join()
    res = sorted(q.get(timeout=5) for _ in range(world))
    bad = sum(r[1] for r in res)
    print(f"world {world}, {ncases} cases: {bad} mismatching (rank, case) pairs; slab-record form in {res[0][2]}, 16-bit records at the end "
          f"of {res[0][3]}, fall-backs to fp32 rows {res[0][4]}; {time.time() - t0:.1f} s")
    sys.exit(1 if bad or any(p.exitcode for p in procs) else 0)
