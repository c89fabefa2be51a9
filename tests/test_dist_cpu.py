"""Multi-process CPU tests (gloo, world_size 2 and 3) of the Z-sharded driver
(euclidean-distance-transform-3d_amd/edt/distributed.py): partitioning, the one-slice label halo
and the Z-slab -> Y-slab all-to-all.  The two local phases are supplied by a CPU implementation
(oracle/edt_oracle.c: oracle_shard_xy / oracle_shard_z) injected through the driver's `ops`
hook, so what is under test is exactly the code that runs unchanged over RCCL on the GPUs."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "euclidean-distance-transform-3d_amd"))


class OracleOps:
    """CPU stand-in for HipOps with the same interface."""

    def __init__(self):
        from oracle import harness
        if not harness.have_port():
            harness.build("port")
        self.lib = harness.port().lib

    def xy(self, labels, halo, code, weights, flags):
        szl, sy, sx = labels.shape
        lab = labels.numpy()
        partial = np.zeros((szl, sy, sx), dtype=np.float32)
        zflags = np.zeros((szl, sy, sx), dtype=np.uint8)
        hp = ctypes.c_void_p(halo.numpy().ctypes.data) if halo is not None else None
        rc = self.lib.oracle_shard_xy(ctypes.c_void_p(lab.ctypes.data), hp, ctypes.c_int(code),
                                      ctypes.c_int64(sx), ctypes.c_int64(sy), ctypes.c_int64(szl),
                                      ctypes.c_float(weights[0]), ctypes.c_float(weights[1]),
                                      ctypes.c_int(flags & 1), ctypes.c_void_p(partial.ctypes.data),
                                      ctypes.c_void_p(zflags.ctypes.data))
        assert rc == 0
        return torch.from_numpy(partial), torch.from_numpy(zflags)

    def z(self, partial, zflags, wz, flags, wxy=None):
        sz, syl, sx = partial.shape
        p = partial.numpy()
        rc = self.lib.oracle_shard_z(ctypes.c_void_p(p.ctypes.data),
                                     ctypes.c_void_p(zflags.numpy().ctypes.data), ctypes.c_int64(sx),
                                     ctypes.c_int64(syl), ctypes.c_int64(sz), ctypes.c_float(wz),
                                     ctypes.c_int(flags & 1), ctypes.c_int(1 if flags & 2 else 0))
        assert rc == 0
        return partial


    # -- slab records (the fast form of the driver), restated with numpy on top of xy / z ------
    def records_supported(self, code, sx, sy, sz):
        return True

    def record_floats(self, sx, ylen):
        return ylen * sx + 2 * (-(-ylen // 32)) * sx

    def xy_records(self, labels, halo, code, weights, flags, y_splits, blocks):
        partial, zflags = self.xy(labels, halo, code, weights, flags)
        p, f = partial.numpy(), zflags.numpy()
        szl, sy, sx = p.shape
        for h, blk in enumerate(blocks):
            ys, ye = y_splits[h], y_splits[h + 1]
            ylen, words = ye - ys, -(-(ye - ys) // 32)
            assert ys % 32 == 0 and tuple(blk.shape) == (szl, self.record_floats(sx, ylen))
            raw = blk.numpy().view(np.uint32)   # shares the tensor's memory
            raw[:, :ylen * sx] = p[:, ys:ye, :].reshape(szl, -1).view(np.uint32)
            bits = np.zeros((szl, 2, words, sx), np.uint32)
            for r in range(ylen):
                row = f[:, ys + r, :].astype(np.uint32)
                bits[:, 0, r // 32, :] |= (row & 1) << (r % 32)
                bits[:, 1, r // 32, :] |= ((row >> 1) & 1) << (r % 32)
            raw[:, ylen * sx:] = bits.reshape(szl, -1)

    def z_records(self, records, sx, syl, wz, flags, wxy=None):
        raw = records.numpy().view(np.uint32)
        sz, words = raw.shape[0], -(-syl // 32)
        partial = raw[:, :syl * sx].view(np.float32).reshape(sz, syl, sx).copy()
        bits = raw[:, syl * sx:].reshape(sz, 2, words, sx)
        zflags = np.zeros((sz, syl, sx), np.uint8)
        for r in range(syl):
            zflags[:, r, :] = ((bits[:, 0, r // 32, :] >> (r % 32)) & 1) | (((bits[:, 1, r // 32, :] >> (r % 32)) & 1) << 1)
        self.z(torch.from_numpy(partial), torch.from_numpy(zflags), wz, flags)
        raw[:, :syl * sx] = partial.reshape(sz, -1).view(np.uint32)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, shape, an, bb, sqrt, gather_back, q, records=None, chunks=None, reuse=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from edt import distributed as edist
        from oracle import harness
        from synth import blocky_labels

        rng = np.random.default_rng(77)
        vol = blocky_labels(shape, nlabels=5, zero_frac=0.15, block=5, rng=rng).astype(np.uint32)
        vol = np.asfortranarray(vol)                      # (sx, sy, sz), x fastest
        zyx = np.ascontiguousarray(vol.T)                  # (sz, sy, sx)
        plan = edist.ShardedEDT(shape, 2, ops=OracleOps(), records=records, chunks=chunks, reuse_output=reuse)
        assert records is None or plan.records == records
        zs, ze = plan.local_z()
        slab = torch.from_numpy(zyx[zs:ze].copy().view(np.int32))
        if reuse:
            # a plan that keeps its receive buffer (and its halo buffer): a first transform of OTHER labels must leave
            # nothing behind that the second one could pick up
            other = torch.from_numpy(np.ascontiguousarray(zyx[zs:ze][:, ::-1, :]).view(np.int32))
            first = plan.run(other, an, black_border=bb, sqrt=sqrt)
            ptr = first.data_ptr()
        out = plan.run(slab, an, black_border=bb, sqrt=sqrt, gather_back=gather_back)
        if reuse and not gather_back:
            assert out.data_ptr() == ptr, "reuse_output: the receive buffer was allocated again"
        out = out.numpy()

        want = harness.port().edtsq(vol, an, bb)
        if sqrt:
            want = np.sqrt(want)
        want = np.ascontiguousarray(want.T)                # (sz, sy, sx)
        if gather_back:
            ok = np.array_equal(out, want[zs:ze], equal_nan=True)
        else:
            ys, ye = plan.local_y()
            ok = np.array_equal(out, want[:, ys:ye, :], equal_nan=True)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,an,bb,sqrt,gather_back", [
    (2, (24, 20, 18), (6.0, 6.0, 30.0), True, False, False),
    (2, (17, 13, 11), (1.0, 1.0, 1.0), False, True, True),
    (3, (16, 19, 22), (0.5, 0.7, 1.3), False, False, False),
    (3, (9, 7, 8), (4.0, 4.0, 40.0), True, False, True),
])
def test_z_sharded_equals_single_process(world, shape, an, bb, sqrt, gather_back):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, an, bb, sqrt, gather_back, q))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0, "worker crashed"
    results = dict(q.get(timeout=5) for _ in range(world))
    assert results == {r: True for r in range(world)}


@pytest.mark.parametrize("world,shape,an,bb,sqrt,gather_back,chunks", [
    (2, (12, 70, 9), (6.0, 6.0, 30.0), True, False, False, 4),     # 3 words of y: 64 + 6 rows
    (2, (10, 64, 7), (1.0, 2.0, 3.0), False, True, True, 2),
    (3, (8, 100, 10), (0.5, 0.7, 1.3), False, False, False, 3),    # 4 words: 64 / 32 / 4 rows
    (3, (9, 97, 3), (4.0, 4.0, 40.0), True, False, True, 1),       # one slice per rank
])
def test_slab_record_form_equals_single_process(world, shape, an, bb, sqrt, gather_back, chunks):
    """The fast form of the driver: y cut at multiples of 32 rows, per-destination records, the
    slab processed in z-chunks whose exchanges are issued before the next chunk's kernels."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, an, bb, sqrt, gather_back, q, True, chunks))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0, "worker crashed"
    results = dict(q.get(timeout=5) for _ in range(world))
    assert results == {r: True for r in range(world)}


@pytest.mark.parametrize("world,shape,chunks", [(2, (12, 70, 9), 4), (3, (8, 100, 10), 3), (2, (10, 64, 6), 1)])
def test_slab_record_form_with_reused_buffers(world, shape, chunks):
    """reuse_output: one receive buffer and one halo buffer per plan; the chunks are taken top-down with the halo exchange
    in flight (waited for before the slab's first chunk only) -- two transforms in a row, the second one checked."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, (1.0, 1.0, 2.0), False, False, False, q, True, chunks, True))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0, "worker crashed"
    results = dict(q.get(timeout=5) for _ in range(world))
    assert results == {r: True for r in range(world)}


def test_partition_helpers():
    from edt.distributed import balanced_partition, global_extents
    assert balanced_partition(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert balanced_partition(8, 8) == [(i, i + 1) for i in range(8)]
    for world in (1, 2, 4, 8):
        e = global_extents(world, 512)
        assert e[0] * e[1] * e[2] == world * 512 ** 3
    assert global_extents(8, 512) == (1024, 1024, 1024)
