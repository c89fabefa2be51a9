"""The WHOLE multi-process driver with the real kernels on ONE GPU (`-m gpu`): world_size 2 and 3 processes
share cuda:0 and talk over gloo, whose point-to-point transfers the driver then stages through host memory
(RCCL refuses two ranks on one device).  Everything except RCCL itself is what runs on an 8-GPU node: the
partition, the one-slice label halo, the chunked slab-record XY phase on two streams, the exchange order,
the Z pass over the gathered records, and the byte-flag fallback form."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, "euclidean-distance-transform-3d_amd"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, shape, an, bb, sqrt, gather_back, records, chunks, q, expect16=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from edt import _lib
        from edt import distributed as edist
        from oracle import harness
        from synth import voronoi_labels

        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        vol = voronoi_labels(shape, nseeds=30, seed=11, upsample=4, membrane=0.04)   # (sx, sy, sz), x fastest
        if expect16 == "fallback":
            vol[:, :, shape[2] // 2:] = 7   # one label over whole slices, 300+ voxels across: beyond 16 bits of quanta at (1, 1, 1)
        zyx = np.ascontiguousarray(vol.T)
        plan = edist.ShardedEDT(shape, _lib.U32, records=records, chunks=chunks)
        assert plan.records == records
        zs, ze = plan.local_z()
        slab = torch.from_numpy(zyx[zs:ze].copy().view(np.int32)).to(dev)
        ok = True
        for _ in range(2):  # twice: the second run reuses streams, send buffers and scratch
            out = plan.run(slab, an, black_border=bb, sqrt=sqrt, gather_back=gather_back).cpu().numpy()
            if not harness.have_port():
                harness.build("port")
            want = harness.port().edtsq(vol, an, bb)
            if sqrt:
                want = np.sqrt(want)
            want = np.ascontiguousarray(want.T)
            if gather_back:
                ok = ok and np.array_equal(out, want[zs:ze], equal_nan=True)
            else:
                ys, ye = plan.local_y()
                ok = ok and np.array_equal(out, want[:, ys:ye, :], equal_nan=True)
        if expect16 == "used":
            ok = ok and plan.last_records16 and plan.fallbacks16 == 0
        elif expect16 == "fallback":
            ok = ok and not plan.last_records16 and plan.fallbacks16 == 1   # (the first run fell back, the second stayed there)
        elif records:
            ok = ok and not plan.last_records16
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,an,bb,sqrt,gather_back,chunks,expect16", [
    (2, (64, 128, 100), (6.0, 6.0, 30.0), True, False, False, 3, "used"),
    (3, (96, 200, 97), (1.0, 1.0, 1.0), True, True, True, 2, "used"),
    (2, (640, 128, 100), (1.0, 1.0, 1.0), True, False, False, 2, "fallback"),
])
def test_processes_sharing_one_gpu_records_of_16_bit_rows(world, shape, an, bb, sqrt, gather_back, chunks, expect16):
    """the driver's 16-bit records with the real kernels: used where every tile has that form; a step that meets tiles beyond
    16 bits is repeated with fp32 records on every rank (one all-reduce of the refused-tile counter) and the plan stays there"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, an, bb, sqrt, gather_back, True, chunks, q, expect16))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, "worker crashed"
    results = dict(q.get(timeout=5) for _ in range(world))
    assert results == {r: True for r in range(world)}


@pytest.mark.parametrize("world,shape,an,bb,sqrt,gather_back,records,chunks", [
    (2, (64, 96, 40), (6.0, 6.0, 30.0), True, False, False, True, 3),
    (2, (128, 200, 17), (1.0, 2.0, 0.5), False, True, True, True, 2),
    (3, (48, 128, 31), (1.0, 1.0, 1.0), False, False, False, True, 4),
    (2, (40, 40, 36), (6.0, 6.0, 30.0), True, False, False, False, None),   # byte-flag form (sy < 64)
])
def test_processes_sharing_one_gpu(world, shape, an, bb, sqrt, gather_back, records, chunks):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, an, bb, sqrt, gather_back, records, chunks, q))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, "worker crashed"
    results = dict(q.get(timeout=5) for _ in range(world))
    assert results == {r: True for r in range(world)}


def test_bench_sharded_leg_on_one_gpu():
    """bench.py's N > 1 leg end to end (torch.distributed.run, 2 ranks sharing cuda:0 over gloo): the cfg4
    segmentation generator per slab, the timed steps, and the bit-for-bit check of the gathered result against
    the compiled reference -- what the driver's 8-GPU run does, minus RCCL."""
    import json
    import subprocess

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, EDT_BENCH_BACKEND="gloo", EDT_BENCH_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "128"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    # (a 128 x 128 x 256 segmentation at (1, 1, 1): both scan axes on the integer kernel, every tile within 16 bits)
    assert out["n_gpus"] == 2 and out["config"]["form"] == "slab records, 16-bit rows" and out["config"]["records16_fallbacks"] == 0
    # two independent steps in flight (own plan and stream each) at N > 1, said in the line
    assert out["config"]["steps_in_flight"] == 2 and "in flight" in out["reading"]
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libedt_ref.so")):
        assert out["config"]["output_verified"] is True, out["config"]
        assert out["cpu_baseline"]["kind"] == "reference"
    # what makes the line readable against N = 1: the kind of scaling, the same workload on one GPU, the efficiency
    assert out["scaling"] == "weak" and "configs[3]" in out["reading"] and out["config"]["labels"] == "cfg4"
    same = out["single_gpu_same_workload"]
    assert same["mvox_per_s"] > 0
    assert out["scaling_efficiency"] == pytest.approx(out["value"] / (2 * same["mvox_per_s"]), abs=1e-4)
    # the self-test ran before the timed steps and says what a failure on a multi-GPU box would be diagnosed from
    st = [ln for ln in res.stdout.splitlines() if ln.startswith("[selftest]")]
    assert any("RCCL/NCCL" in ln for ln in st) and any("peer access" in ln for ln in st)
    assert any("all_to_all of rank 0" in ln for ln in st)
    assert any("output_verified=True" in ln for ln in st), st
    assert res.stdout.strip().splitlines()[-1].startswith("{"), "the JSON line must be the last line of stdout"


@pytest.mark.parametrize("extra,scaling,labels", [(["--labels", "ones", "--pipeline", "1"], "weak", "ones"),
                                                  (["--global-size", "128"], "strong", "cfg4"),
                                                  (["--global-size", "128", "--labels", "ones"], "strong", "ones")])
def test_bench_sharded_leg_other_readings(extra, scaling, labels):
    """--labels ones: the series that continues the N = 1 headline (closed-form check); --global-size: strong scaling."""
    import json
    import subprocess

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, EDT_BENCH_BACKEND="gloo", EDT_BENCH_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "128", "--no-selftest"] + extra
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["scaling"] == scaling and out["config"]["labels"] == labels
    assert out["config"]["global_extents"] == ([128, 128, 128] if scaling == "strong" else [128, 128, 256])
    assert out["scaling_efficiency"] == pytest.approx(out["value"] / (2 * out["single_gpu_same_workload"]["mvox_per_s"]), abs=1e-4)
    if labels == "ones" or os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libedt_ref.so")):
        assert out["config"]["output_verified"] is True, out["config"]


def _nccl_worker(rank, world, port, shape, an, bb, chunks, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from edt import _lib
        from edt import distributed as edist
        from oracle import harness
        from synth import voronoi_labels

        vol = voronoi_labels(shape, nseeds=40, seed=5, upsample=4, membrane=0.03)   # (sx, sy, sz), x fastest
        zyx = np.ascontiguousarray(vol.T)
        plan = edist.ShardedEDT(shape, _lib.U32, chunks=chunks)
        zs, ze = plan.local_z()
        ys, ye = plan.local_y()
        slab = torch.from_numpy(zyx[zs:ze].copy().view(np.int32)).to(dev)
        chk = harness.ref() if harness.have_ref() else harness.port()
        want = np.ascontiguousarray(chk.edtsq(vol, an, bb).T)
        ok = True
        for _ in range(2):
            out = plan.run(slab, an, black_border=bb).cpu().numpy()
            ok = ok and np.array_equal(out, want[:, ys:ye, :], equal_nan=True)
            back = plan.run(slab, an, black_border=bb, gather_back=True).cpu().numpy()
            ok = ok and np.array_equal(back, want[zs:ze], equal_nan=True)
        q.put((rank, bool(ok), bool(plan.records)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4, 8])
@pytest.mark.parametrize("shape,an,bb,chunks", [
    ((128, 320, 96), (1.0, 1.0, 1.0), False, 3),      # slab records, chunked: the exchange of one chunk under the next one's kernels
    ((96, 290, 70), (6.0, 6.0, 30.0), True, 1),       # uneven cuts (sy, sz not multiples of the world size), one chunk
    ((256, 250, 254), (1.0, 1.0, 1.0), False, 4),     # a 1024 x 1000 x 1016-style volume at a quarter of the size: uneven everywhere
])
def test_rccl_exchange_between_real_devices(world, shape, an, bb, chunks):
    """The one path no 1-GPU box can run: RCCL `all_to_all` of slab records between DIFFERENT devices (non-empty
    peers), the one-slice label halo over send/recv (in flight while the upper chunks run) and the chunked overlap, checked
    bit for bit against the compiled reference, at every world size the box offers up to 8 -- skipped where the box has
    fewer devices (gpurun boxes have one); the driver's first multi-GPU lease runs it."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < world:
        pytest.skip(f"needs >= {world} GPUs (RCCL refuses two ranks on one device)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_worker, args=(r, world, port, shape, an, bb, chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, "worker crashed"
    results = [q.get(timeout=5) for _ in range(world)]
    assert sorted(r[0] for r in results) == list(range(world))
    assert all(r[1] for r in results), results
    assert all(r[2] for r in results), "expected the slab-record form"
