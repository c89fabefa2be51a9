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


    # -- slab records of 16-bit values: the same restatement, rows as integers in quanta (edt_hip.h) ------------------------
    allow16 = True

    @staticmethod
    def _quantum(weights):
        """q with w_i^2 = a_i q for integer a_i (integer voxel sizes only: all the tests need)"""
        sq = [int(round(float(w) ** 2)) for w in weights]
        if any(abs(float(w) ** 2 - v) > 0 or v <= 0 for w, v in zip(weights, sq)):
            return None
        return float(np.gcd.reduce(sq))

    def records16_supported(self, code, sx, sy, sz, weights):
        return self.allow16 and sx % 4 == 0 and self._quantum(weights) is not None

    def record16_words(self, sx, ylen):
        return ylen * sx // 2 + 2 * (-(-ylen // 32)) * sx

    def xy_records16(self, labels, halo, code, weights, flags, y_splits, blocks, refused):
        partial, zflags = self.xy(labels, halo, code, weights, flags)
        q = self._quantum(weights)
        p, f = partial.numpy(), zflags.numpy()
        szl, sy, sx = p.shape
        n = p / np.float32(q)
        bad = ~np.isfinite(n) | (n != np.floor(n)) | (n > 65535)
        # a 32-column tile of one slice with any such voxel has no 16-bit form: counted, its rows unspecified (here: garbage)
        tiles = bad.reshape(szl, sy, -1, min(32, sx)).any(axis=(1, 3)) if sx % 32 == 0 else bad.any(axis=1, keepdims=True)
        refused += int(tiles.sum())
        n16 = np.where(bad, 0xDEAD, n).astype(np.uint16)
        for h, blk in enumerate(blocks):
            ys, ye = y_splits[h], y_splits[h + 1]
            ylen, words = ye - ys, -(-(ye - ys) // 32)
            assert ys % 32 == 0 and tuple(blk.shape) == (szl, self.record16_words(sx, ylen)) and blk.dtype == torch.int32
            raw = blk.numpy().view(np.uint32)
            raw[:, :ylen * sx // 2] = np.ascontiguousarray(n16[:, ys:ye, :]).reshape(szl, -1).view(np.uint32)
            bits = np.zeros((szl, 2, words, sx), np.uint32)
            for r in range(ylen):
                row = f[:, ys + r, :].astype(np.uint32)
                bits[:, 0, r // 32, :] |= (row & 1) << (r % 32)
                bits[:, 1, r // 32, :] |= ((row >> 1) & 1) << (r % 32)
            raw[:, ylen * sx // 2:] = bits.reshape(szl, -1)

    def z_records16(self, records, out, weights, flags):
        sz, syl, sx = out.shape
        q = self._quantum(weights)
        raw = records.numpy().view(np.uint32)
        words = -(-syl // 32)
        n16 = np.ascontiguousarray(raw[:, :syl * sx // 2]).view(np.uint16).reshape(sz, syl, sx)
        partial = (n16.astype(np.float32) * np.float32(q)).copy()
        bits = raw[:, syl * sx // 2:].reshape(sz, 2, words, sx)
        zflags = np.zeros((sz, syl, sx), np.uint8)
        for r in range(syl):
            zflags[:, r, :] = ((bits[:, 0, r // 32, :] >> (r % 32)) & 1) | (((bits[:, 1, r // 32, :] >> (r % 32)) & 1) << 1)
        self.z(torch.from_numpy(partial), torch.from_numpy(zflags), weights[2], flags)
        out.copy_(torch.from_numpy(partial))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, shape, an, bb, sqrt, gather_back, q, records=None, chunks=None, reuse=False, expect16=None,
            deep=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from edt import distributed as edist
        from oracle import harness
        from synth import blocky_labels

        rng = np.random.default_rng(77)
        vol = blocky_labels(shape, nlabels=5, zero_frac=0.15, block=5, rng=rng).astype(np.uint32)
        if deep:
            vol[:, :, shape[2] // 2:] = 3                 # one label over whole slices: rows deeper than 16 bits of quanta hold
        vol = np.asfortranarray(vol)                      # (sx, sy, sz), x fastest
        zyx = np.ascontiguousarray(vol.T)                  # (sz, sy, sx)
        ops = OracleOps()
        ops.allow16 = expect16 is not None
        plan = edist.ShardedEDT(shape, 2, ops=ops, records=records, chunks=chunks, reuse_output=reuse)
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
        if expect16 == "used":
            ok = ok and plan.last_records16 and plan.fallbacks16 == 0
        elif expect16 == "fallback":
            # the step met tiles without a 16-bit form: every rank repeated it with fp32 records, and the plan stays there
            ok = ok and not plan.last_records16 and plan.fallbacks16 == 1
            again = plan.run(slab, an, black_border=bb, sqrt=sqrt, gather_back=gather_back).numpy()
            ok = ok and np.array_equal(again, out, equal_nan=True) and plan.fallbacks16 == 1
            # ... for a while: the next attempt comes 8 steps later (then 16, 32, ...), or at once after reset_records16()
            ok = ok and plan._backoff16 == 7 and plan._backoff16_next == 16
            plan.reset_records16()
            third = plan.run(slab, an, black_border=bb, sqrt=sqrt, gather_back=gather_back).numpy()
            ok = ok and np.array_equal(third, out, equal_nan=True) and plan.fallbacks16 == 2 and not plan.last_records16
            ok = ok and plan._backoff16 == 8 and plan._backoff16_next == 16
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
    (2, (8, 64, 41), (1.0, 1.0, 2.0), True, False, False, 4),      # 20 / 21 slices per rank: chunk 0 (processed last) is half a share
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


@pytest.mark.parametrize("world,shape,an,bb,sqrt,gather_back,chunks,expect16,deep", [
    (2, (32, 70, 9), (6.0, 6.0, 30.0), True, False, False, 2, "used", False),
    (3, (64, 100, 10), (1.0, 2.0, 3.0), True, True, True, 3, "used", False),
    (2, (12, 64, 7), (2.0, 2.0, 2.0), True, False, False, 1, "used", False),
    (2, (128, 128, 9), (30.0, 30.0, 6.0), True, False, False, 2, "fallback", True),   # 128 x 128 of one label: 25 k^2 quanta > 2^16
    (3, (128, 160, 10), (30.0, 30.0, 6.0), True, False, True, 3, "fallback", True),
    (2, (32, 70, 9), (1.0, 1.0, 1.0), False, False, False, 2, "fallback", True),    # no black border: rows without a boundary
])
def test_slab_records_of_16_bit_rows(world, shape, an, bb, sqrt, gather_back, chunks, expect16, deep):
    """Records of 16-bit rows (2.25 bytes per voxel) where the voxel sizes share a quantum: used when every tile has that
    form; when some rank meets a tile that has not, all ranks learn it from one all-reduce and repeat the step with fp32
    records -- the driver logic of edt/distributed.py over gloo, the phases restated on the CPU."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, an, bb, sqrt, gather_back, q, True, chunks, False,
                                               expect16, deep)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0, "worker crashed"
    results = dict(q.get(timeout=5) for _ in range(world))
    assert results == {r: True for r in range(world)}


def _random_worker(rank, world, port, ncases, seed, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from edt import distributed as edist
        from oracle import harness
        from synth import blocky_labels

        rng = np.random.default_rng(seed)   # (the same cases on every rank)
        port_lib = harness.port()
        bad, n16, nfall = 0, 0, 0
        for _ in range(ncases):
            sx = 4 * int(rng.integers(1, 12)) if rng.random() < 0.8 else int(rng.integers(3, 40))
            sy = int(rng.integers(32 * world, 32 * world + 80))
            sz = int(rng.integers(world, 30))
            an = tuple(float(a) for a in rng.choice([1, 2, 6, 30, 4, 3], size=3))
            bb, sqrt, gather_back = bool(rng.integers(0, 2)), rng.random() < 0.3, rng.random() < 0.3
            chunks = int(rng.integers(1, 5))
            vol = blocky_labels((sx, sy, sz), nlabels=int(rng.integers(1, 9)), zero_frac=float(rng.random() * 0.2),
                                block=int(rng.integers(2, 40)), rng=rng).astype(np.uint32)
            if rng.random() < 0.2:
                vol[:] = 1
            vol = np.asfortranarray(vol)
            zyx = np.ascontiguousarray(vol.T)
            ops = OracleOps()
            ops.allow16 = bool(rng.integers(0, 2))
            plan = edist.ShardedEDT((sx, sy, sz), 2, ops=ops, chunks=chunks, reuse_output=bool(rng.integers(0, 2)))
            zs, ze = plan.local_z()
            slab = torch.from_numpy(zyx[zs:ze].copy().view(np.int32))
            want = port_lib.edtsq(vol, an, bb)
            if sqrt:
                want = np.sqrt(want)
            want = np.ascontiguousarray(want.T)
            for _rep in range(2):
                out = plan.run(slab, an, black_border=bb, sqrt=sqrt, gather_back=gather_back).numpy()
                if gather_back:
                    ok = np.array_equal(out, want[zs:ze], equal_nan=True)
                else:
                    ys, ye = plan.local_y()
                    ok = np.array_equal(out, want[:, ys:ye, :], equal_nan=True)
                bad += not ok
            n16 += plan.last_records16
            nfall += plan.fallbacks16
        q.put((rank, bad, n16, nfall))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,ncases,seed", [(2, 14, 5), (3, 10, 6)])
def test_driver_random_cases_over_gloo(world, ncases, seed):
    """the driver walked through random small cases (extents, voxel sizes, chunks, gather-back, reused buffers, 16-bit records
    allowed or not -- with steps that fall back), every case run twice, the phases restated on the CPU: the CPU tier's
    counterpart of tools/fuzz_driver.py"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_random_worker, args=(r, world, port, ncases, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, "worker crashed"
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert all(r[1] == 0 for r in res), res
    assert all(r[2:] == res[0][2:] for r in res), "the ranks disagree about the form of the records"


def test_chunks_cover_the_slab_and_the_last_processed_one_is_small():
    from edt.distributed import ShardedEDT

    class Plan(ShardedEDT):
        def __init__(self, zparts, nchunks):
            self.zparts, self.nchunks, self.world = zparts, nchunks, len(zparts)

    for zparts, nchunks in (([(0, 128), (128, 256)], 4), ([(0, 20), (20, 41)], 4), ([(0, 5), (5, 9)], 4), ([(0, 64)], 1), ([(0, 7), (7, 15)], 3)):
        p = Plan(zparts, nchunks)
        for r, (zs, ze) in enumerate(zparts):
            cuts = [p._chunk(r, k) for k in range(nchunks)]
            assert cuts[0][0] == zs and cuts[-1][1] == ze and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            assert all(c1 > c0 for c0, c1 in cuts)
            if nchunks >= 2 and ze - zs >= 2 * nchunks and len(zparts) > 1:
                assert cuts[0][1] - cuts[0][0] == (ze - zs) // (2 * nchunks)   # e.g. 16 of 128 slices at 4 chunks


def test_partition_helpers():
    from edt.distributed import balanced_partition, global_extents
    assert balanced_partition(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert balanced_partition(8, 8) == [(i, i + 1) for i in range(8)]
    for world in (1, 2, 4, 8):
        e = global_extents(world, 512)
        assert e[0] * e[1] * e[2] == world * 512 ** 3
    assert global_extents(8, 512) == (1024, 1024, 1024)
