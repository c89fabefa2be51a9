"""Z-sharded multi-GPU EDT: one process per GPU, torch.distributed (RCCL over xGMI).

Why this shape ("Multi-GPU"): the X and Y passes of the reference touch one
z-slice at a time (reference: src/edt.hpp:430-460), so a volume cut into contiguous Z-slabs
needs no communication for them.  The Z pass (src/edt.hpp:465-475) walks whole z-columns, which
a fixed-width halo cannot provide exactly, so the fp32 partial result is re-partitioned ONCE
from Z-slabs to Y-slabs with an all-to-all; every rank then owns complete z-columns for its
y-range.  The all-to-all is issued as one group of point-to-point sends/receives -- on a
fully connected xGMI node every peer pair has its own link, so the 7 transfers of a rank run
concurrently.  Labels do not travel: one flag byte per voxel (foreground, run-start-along-z)
does, and run continuity across the slab cut is decided from a ONE-SLICE label halo received
from the previous rank.

    rank r:  labels[z in Z_r, :, :] --X,Y passes--> partial, zflags        (local, HIP)
             halo: last slice of rank r-1's labels                          (1 send/recv)
             all-to-all: block (Z_r, Y_h) -> rank h                         (P2P group)
             Z pass on [all z, y in Y_r, :]                                 (local, HIP)

The numerical work is done by `ops` (default: the HIP kernels through the C ABI); the
partition / exchange logic here is backend agnostic, which is how the CPU test-suite drives
it over gloo with a CPU implementation of the two phases.

Fast form ("slab records", include/edt_hip.h): when the extents allow it (sx, sy, sz <= 2048, at least
one 32-row word of y per rank) the y axis is cut at multiples of 32 rows and the XY phase writes,
for every destination rank, one contiguous record per xy-slice -- that rank's rows after the
X and Y passes followed by their foreground / z-run-start BITS (4.25 bytes per voxel).  A
peer's message is then contiguous on both sides (no pack or unpack copies; the rank's own part is
written straight into its receive buffer), and the slab is processed in z-chunks so that the
exchange of chunk k runs (on RCCL's stream) under the kernels of chunk k+1:

    for chunk k:   XY kernels(chunk k) -> records            (compute stream)
                   isend/irecv group(chunk k), not waited    (communication stream)
    wait all;      Z pass over the gathered records          (compute stream)
"""
from __future__ import annotations

import ctypes
import json
import time

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


def balanced_partition(n: int, parts: int):
    """`parts` contiguous ranges covering [0, n), sizes differing by at most one."""
    base, extra = divmod(n, parts)
    out, start = [], 0
    for i in range(parts):
        size = base + (1 if i < extra else 0)
        out.append((start, start + size))
        start += size
    return out


class HipOps:
    """The two local phases on the GPU (edt_hip_shard_xy_device / edt_hip_shard_z_device)."""

    def __init__(self):
        self.lib = _lib.load()
        self._ws = {}


    def _workspace(self, nbytes, device, slot=0):
        """Scratch of one stream of work (`slot`): calls that may overlap use different slots."""
        ws = self._ws.get(slot)
        if ws is None or ws.numel() < nbytes or ws.device != device:
            ws = self._ws[slot] = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        return ws

    @staticmethod
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def xy(self, labels, halo, code, weights, flags):
        szl, sy, sx = labels.shape
        partial = torch.empty((szl, sy, sx), dtype=torch.float32, device=labels.device)
        zflags = torch.empty((szl, sy, sx), dtype=torch.uint8, device=labels.device)
        ws = self._workspace(self.lib.edt_hip_shard_workspace_bytes(code, sx, sy, szl), labels.device)
        _lib.check(self.lib.edt_hip_shard_xy_device(
            ctypes.c_void_p(labels.data_ptr()),
            ctypes.c_void_p(halo.data_ptr()) if halo is not None else None, code, sx, sy, szl,
            weights[0], weights[1], flags, ctypes.c_void_p(partial.data_ptr()),
            ctypes.c_void_p(zflags.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ws.numel(),
            self._stream()))
        return partial, zflags

    # -- slab records (fast form) -------------------------------------------------------------
    def records_supported(self, code, sx, sy, sz):
        return bool(self.lib.edt_hip_shard_records_supported(code, sx, sy, sz))

    def record_floats(self, sx, ylen):
        return int(self.lib.edt_hip_shard_record_floats(sx, ylen))

    def xy_records(self, labels, halo, code, weights, flags, y_splits, blocks, slot=0):
        """X and Y passes of a z-chunk; block h receives chunk-many records of destination h."""
        szl, sy, sx = labels.shape
        ws = self._workspace(self.lib.edt_hip_shard_records_workspace_bytes(code, sx, sy, szl), labels.device,
                             slot)
        splits = (ctypes.c_int64 * len(y_splits))(*y_splits)
        ptrs = (ctypes.c_void_p * len(blocks))(*[b.data_ptr() for b in blocks])
        _lib.check(self.lib.edt_hip_shard_xy_records_device(
            ctypes.c_void_p(labels.data_ptr()),
            ctypes.c_void_p(halo.data_ptr()) if halo is not None else None, code, sx, sy, szl,
            weights[0], weights[1], flags, len(blocks), splits, ptrs, ctypes.c_void_p(ws.data_ptr()),
            ws.numel(), self._stream()))

    def z_records(self, records, sx, syl, wz, flags, wxy=None):
        """Z pass in place over the gathered (sz, record_floats) buffer.  wxy = (wx, wy) of the XY phase that produced the
        records, where the caller knows them (an argument of THIS call, never remembered): the Z pass may then run on the
        integer column kernel and form fp32 candidates -- the same bits either way (edt_hip.h)."""
        sz = records.shape[0]
        ws = self._workspace(self.lib.edt_hip_shard_records_workspace_bytes(_lib.U8, sx, syl, sz), records.device)
        if wxy is not None:
            _lib.check(self.lib.edt_hip_shard_z_records_device_w(
                ctypes.c_void_p(records.data_ptr()), sx, syl, sz, wxy[0], wxy[1], wz, flags, ctypes.c_void_p(ws.data_ptr()),
                ws.numel(), self._stream()))
        else:
            _lib.check(self.lib.edt_hip_shard_z_records_device_ex(
                ctypes.c_void_p(records.data_ptr()), sx, syl, sz, wz, 0.0, flags, ctypes.c_void_p(ws.data_ptr()),
                ws.numel(), self._stream()))

    # -- slab records of 16-bit values (edt_hip.h): 2.25 bytes per voxel where the integer column kernel takes both passes --
    def records16_supported(self, code, sx, sy, sz, weights):
        return bool(self.lib.edt_hip_shard_records16_supported(code, sx, sy, sz, weights[0], weights[1], weights[2]))

    def record16_words(self, sx, ylen):
        return int(self.lib.edt_hip_shard_record16_words(sx, ylen))

    def xy_records16(self, labels, halo, code, weights, flags, y_splits, blocks, refused, slot=0):
        """X and Y passes of a z-chunk into records of 16-bit rows (int32 tensors of record16_words columns); `refused`
        (int32[1], zeroed by the caller) counts the tiles that have no 16-bit form."""
        szl, sy, sx = labels.shape
        ws = self._workspace(self.lib.edt_hip_shard_records_workspace_bytes(code, sx, sy, szl), labels.device, slot)
        splits = (ctypes.c_int64 * len(y_splits))(*y_splits)
        ptrs = (ctypes.c_void_p * len(blocks))(*[b.data_ptr() for b in blocks])
        _lib.check(self.lib.edt_hip_shard_xy_records16_device(
            ctypes.c_void_p(labels.data_ptr()),
            ctypes.c_void_p(halo.data_ptr()) if halo is not None else None, code, sx, sy, szl,
            weights[0], weights[1], weights[2], flags, len(blocks), splits, ptrs, ctypes.c_void_p(refused.data_ptr()),
            ctypes.c_void_p(ws.data_ptr()), ws.numel(), self._stream()))

    def z_records16(self, records, out, weights, flags):
        """Z pass over the gathered (sz, record16_words) records into the dense (sz, syl, sx) fp32 tensor `out`."""
        sz, syl, sx = out.shape
        ws = self._workspace(self.lib.edt_hip_shard_records_workspace_bytes(_lib.U8, sx, syl, sz), records.device)
        _lib.check(self.lib.edt_hip_shard_z_records16_device(
            ctypes.c_void_p(records.data_ptr()), ctypes.c_void_p(out.data_ptr()), sx, syl, sz, weights[0], weights[1],
            weights[2], flags, ctypes.c_void_p(ws.data_ptr()), ws.numel(), self._stream()))

    def z(self, partial, zflags, wz, flags, wxy=None):
        sz, syl, sx = partial.shape
        ws = self._workspace(self.lib.edt_hip_shard_workspace_bytes(_lib.U8, sx, syl, sz), partial.device)
        floor = float(self.lib.edt_hip_field_floor(wxy[0], wxy[1])) if wxy is not None else 0.0
        _lib.check(self.lib.edt_hip_shard_z_device_ex(
            ctypes.c_void_p(partial.data_ptr()), ctypes.c_void_p(zflags.data_ptr()), sx, syl, sz, wz, floor,
            flags, ctypes.c_void_p(ws.data_ptr()), ws.numel(), self._stream()))
        return partial


class _Transfers:
    """Point-to-point transfers in flight (+ the copies that finish host-staged receives)."""

    def __init__(self, reqs, post):
        self.reqs, self.post = reqs, post

    def wait(self):
        for r in self.reqs:
            r.wait()
        for dst, buf in self.post:
            dst.copy_(buf)
        self.reqs, self.post = [], []


class ShardedEDT:
    """Distributed squared EDT of one volume of x-fastest extents ``(sx, sy, sz)``.

    Every rank passes its Z-slab as a contiguous tensor of shape ``(sz_local, sy, sx)`` and
    receives its Y-slab of the result, shape ``(sz, sy_local, sx)`` (all z, its y-range), or --
    with ``gather_back=True`` -- its original Z-slab of the result.
    """

    def __init__(self, extents_xyz, code: int, group=None, ops=None, records=None, chunks=None, reuse_output=False,
                 records16=None):
        """records: None = use the slab-record form whenever it applies, False = never (the
        byte-flag form).  chunks: z-chunks per slab in the record form (default 4 when world > 1).
        reuse_output: the record form keeps ONE receive buffer between calls -- the view run() returns is then only valid
        until the next run() of this plan (what a loop that consumes every result wants: no allocation per step)."""
        # records16: None = 16-bit records whenever the extents and the voxel sizes of a run() allow them (edt_hip.h; a run
        # that meets a tile without a 16-bit form is repeated with fp32 records and the plan stays on those), False = never
        # (EDT_SHARD_RECORDS16=0 likewise).  `last_records16` says what the last run() used.
        import os as _os
        self._allow16 = records16 is not False and _os.environ.get("EDT_SHARD_RECORDS16", "1") != "0"
        self.last_records16 = False
        self.fallbacks16 = 0
        # After a step that had to be repeated with fp32 records the plan stays on those for `_backoff16` steps, then tries
        # the 16-bit form again; every further fall-back doubles the wait (8, 16, ... 1024 steps): data that refuses every
        # time pays a wasted XY phase ever more rarely, data that refused once (one volume of a stream) gets the 1.9x back.
        # All ranks count alike (the verdict is an all-reduce), so they switch in the same step.  reset_records16() re-arms
        # at once; retry16 = False restores round 4's "stays on fp32 for good".
        self.retry16 = True
        self._backoff16 = 0      # fp32 steps still to run before the next 16-bit attempt
        self._backoff16_next = 8
        self._dst16 = None
        self._out16 = None
        self._refused = None
        self._refused_host = None
        self.reuse_output = bool(reuse_output)
        self._dst = None
        self._halo_buf = None
        self.sx, self.sy, self.sz = (int(e) for e in extents_xyz)
        self.code = code
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        if self.sz < self.world or self.sy < self.world:
            raise ValueError("need at least one z-slice and one y-row per rank")
        self.zparts = balanced_partition(self.sz, self.world)
        self.ops = HipOps() if ops is None else ops
        self._stage = dist.get_backend(group) == "gloo"   # device tensors travel through host copies
        words = -(-self.sy // 32)
        can = (records is not False and hasattr(self.ops, "xy_records") and words >= self.world
               and self.ops.records_supported(code, self.sx, self.sy, self.sz))
        if records is True and not can:
            raise ValueError("the slab-record form does not apply to these extents")
        self.records = bool(can)
        if self.records:
            # y is cut at multiples of 32 rows: a bit word never straddles two ranks
            self.yparts = [(32 * a, min(32 * b, self.sy)) for a, b in balanced_partition(words, self.world)]
            import os
            if chunks is None and os.environ.get("EDT_SHARD_CHUNKS"):
                chunks = int(os.environ["EDT_SHARD_CHUNKS"])
            want = chunks if chunks is not None else (4 if self.world > 1 else 1)
            self.nchunks = max(1, min(int(want), min(e - s for s, e in self.zparts)))
            self._send = {}
            self._streams = None
            self._nstreams = int(os.environ.get("EDT_SHARD_STREAMS", "2"))
            # how a chunk travels: "alltoall" = one dist.all_to_all per chunk (RCCL), "p2p" = a batch
            # of isend / irecv (any backend; gloo has no list all_to_all).  EDT_SHARD_EXCHANGE overrides.
            default = "alltoall" if dist.get_backend(group) == "nccl" else "p2p"
            self._exchange = os.environ.get("EDT_SHARD_EXCHANGE", default)
            if self._exchange not in ("alltoall", "p2p"):
                raise ValueError("EDT_SHARD_EXCHANGE must be 'alltoall' or 'p2p'")
        else:
            self.yparts = balanced_partition(self.sy, self.world)

    # -- helpers ----------------------------------------------------------------------------
    def local_z(self):
        return self.zparts[self.rank]

    def local_y(self):
        return self.yparts[self.rank]

    def _global_rank(self, r):
        return r if self.group is None else dist.get_global_rank(self.group, r)

    def _p2p(self, sends, recvs):
        """One group of point-to-point transfers: sends / recvs are lists of (tensor, peer rank).
        A backend that cannot move device memory (gloo) is served through host copies -- slow, but it
        lets the whole multi-process driver run with the real kernels on a single GPU (tests)."""
        ops, post = [], []
        for t, peer in sends:
            buf = t.contiguous()
            if self._stage and buf.is_cuda:
                buf = buf.cpu()
            ops.append(dist.P2POp(dist.isend, buf, self._global_rank(peer), self.group))
        for t, peer in recvs:
            buf = t
            if self._stage and t.is_cuda:
                buf = torch.empty(t.shape, dtype=t.dtype, device="cpu")
                post.append((t, buf))
            ops.append(dist.P2POp(dist.irecv, buf, self._global_rank(peer), self.group))
        return _Transfers(dist.batch_isend_irecv(ops) if ops else [], post)

    def _halo_start(self, labels):
        """One-slice label halo: receive the previous rank's last slice, send ours onward.  Enqueue only: returns
        (halo buffer | None, transfers to wait for before the buffer is read)."""
        sends, recvs, halo = [], [], None
        if self.rank + 1 < self.world:
            sends.append((labels[-1], self.rank + 1))
        if self.rank > 0:
            halo = self._halo_buf
            if halo is None or halo.shape != labels[0].shape or halo.dtype != labels.dtype or halo.device != labels.device:
                halo = self._halo_buf = torch.empty_like(labels[0])
            recvs.append((halo, self.rank - 1))
        return halo, self._p2p(sends, recvs)

    def _halo(self, labels):
        halo, req = self._halo_start(labels)
        req.wait()
        return halo

    def _reshard(self, slab_list, to_y: bool):
        """All-to-all between the Z-slab and the Y-slab layouts, as one group of P2P transfers.

        to_y=True : [(szl, sy, sx)]  -> [(sz, syl, sx)]     block (Z_me, Y_h) goes to rank h
        to_y=False: [(sz, syl, sx)]  -> [(szl, sy, sx)]     the inverse
        """
        ys, ye = self.yparts[self.rank]
        zs, ze = self.zparts[self.rank]
        outs, sends, recvs, keep = [], [], [], []
        for src in slab_list:
            if to_y:
                dst = torch.empty((self.sz, ye - ys, self.sx), dtype=src.dtype, device=src.device)
            else:
                dst = torch.empty((ze - zs, self.sy, self.sx), dtype=src.dtype, device=src.device)
            outs.append(dst)
            for h in range(self.world):
                hys, hye = self.yparts[h]
                hzs, hze = self.zparts[h]
                if to_y:
                    send = src[:, hys:hye, :]
                    recv = dst[hzs:hze]                      # contiguous: z is the slowest axis
                else:
                    send = src[hzs:hze]
                    recv = None                              # strided target: stage then copy
                if h == self.rank:
                    if to_y:
                        recv.copy_(send)
                    else:
                        dst[:, hys:hye, :].copy_(send)
                    continue
                sends.append((send, h))
                if to_y:
                    recvs.append((recv, h))
                else:
                    stage = torch.empty((ze - zs, hye - hys, self.sx), dtype=src.dtype, device=src.device)
                    keep.append((stage, dst, hys, hye))
                    recvs.append((stage, h))
        self._p2p(sends, recvs).wait()
        for stage, dst, hys, hye in keep:
            dst[:, hys:hye, :].copy_(stage)
        return outs

    def _chunk(self, r, k):
        """z-range (global indices) of chunk k of rank r's slab; the same rule on every rank.  The chunks are processed
        top-down and only the LAST one's exchange has nothing to hide under: chunk 0 -- the last one processed -- is half
        a fair share, the others split the rest evenly."""
        zs, ze = self.zparts[r]
        n = ze - zs
        if self.nchunks >= 2 and n >= 2 * self.nchunks and self.world > 1:  # (one rank: nothing travels, nothing to hide)
            first = max(1, n // (2 * self.nchunks))
            if k == 0:
                return zs, zs + first
            c0, c1 = balanced_partition(n - first, self.nchunks - 1)[k - 1]
            return zs + first + c0, zs + first + c1
        c0, c1 = balanced_partition(n, self.nchunks)[k]
        return zs + c0, zs + c1

    def _use16(self, w):
        if not (self._allow16 and hasattr(self.ops, "xy_records16")
                and self.ops.records16_supported(self.code, self.sx, self.sy, self.sz, w)):
            return False
        if self._backoff16 > 0:       # (a fall-back not long ago: fp32 records for now)
            self._backoff16 -= 1
            return False
        return True

    def reset_records16(self):
        """Try records of 16-bit rows again from the next run() on (after a fall-back: see `retry16`)."""
        self._backoff16 = 0
        self._backoff16_next = 8

    def _run_records(self, labels, w, flags, sqrt, halo, halo_req=None, use16=False, halo_start=None):
        """Slab-record form: chunked XY phase with the exchange of one chunk under the kernels of the next.  The chunks are
        taken TOP-DOWN: only the slab's first chunk needs the neighbour's slice (halo, in flight: halo_req), and it runs
        last -- the halo exchange is off the critical path whenever there is more than one chunk.
        use16: records of 16-bit rows (int32 words); the ranks agree afterwards whether every tile had that form.
        halo_start: the halo exchange has not been posted yet -- it is, right after the first chunk's kernels and exchange
        have been enqueued (nothing of it is needed before the last chunk; the host reaches its first launch sooner)."""
        zs, ze = self.local_z()
        ys, ye = self.local_y()
        if use16:
            rec = [self.ops.record16_words(self.sx, b - a) for a, b in self.yparts]
            rdtype = torch.int32
        else:
            rec = [self.ops.record_floats(self.sx, b - a) for a, b in self.yparts]
            rdtype = torch.float32
        y_splits = [a for a, _ in self.yparts] + [self.sy]
        dst = (self._dst16 if use16 else self._dst) if self.reuse_output else None
        if dst is None or tuple(dst.shape) != (self.sz, rec[self.rank]) or dst.device != labels.device:
            dst = torch.empty((self.sz, rec[self.rank]), dtype=rdtype, device=labels.device)
            if self.reuse_output and use16:
                self._dst16 = dst
            elif self.reuse_output:
                self._dst = dst
        if use16:
            if self._refused is None or self._refused.device != labels.device:
                self._refused = torch.zeros(1, dtype=torch.int32, device=labels.device)
                self._refused_host = torch.zeros(1, dtype=torch.int32)
                if labels.is_cuda:
                    self._refused_host = self._refused_host.pin_memory()
            self._refused.zero_()
        pending = []
        # On the GPU consecutive chunks alternate between two side streams (each with its own scratch):
        # pass 1 of chunk k+1 fills the tail of chunk k's Y pass, and every exchange is ordered after
        # exactly the kernels that produced its blocks.
        side = None
        if labels.is_cuda and self.nchunks > 1 and self._nstreams > 1:
            main = torch.cuda.current_stream(labels.device)
            if self._streams is None:
                self._streams = [torch.cuda.Stream(labels.device) for _ in range(self._nstreams)]
            side = self._streams
            for st in side:
                st.wait_stream(main)
        if halo_start is not None and self.nchunks == 1:
            halo, halo_req = halo_start()
            halo_start = None
        for i, k in enumerate(reversed(range(self.nchunks))):
            if i == 1 and halo_start is not None:
                halo, halo_req = halo_start()
            # (a later chunk continues this slab: the slice below it; the first one: the neighbour's slice, which has had
            # the other chunks' kernels to arrive -- waited for on the stream that reads it)
            h = halo if k == 0 else labels[self._chunk(self.rank, k)[0] - zs - 1]
            if side is not None:
                with torch.cuda.stream(side[i % len(side)]):
                    if k == 0 and halo_req is not None:
                        halo_req.wait()
                    self._records_chunk(k, labels, h, w, flags, rec, y_splits, dst, pending, i % len(side), use16)
            else:
                if k == 0 and halo_req is not None:
                    halo_req.wait()
                self._records_chunk(k, labels, h, w, flags, rec, y_splits, dst, pending, None, use16)
        if side is not None:
            for st in side:
                main.wait_stream(st)
        agreed = None
        if use16:
            # every rank learns whether ANY rank met a tile without a 16-bit form: one 4-byte all-reduce behind the last
            # exchange, read back on a side stream while the Z phase runs -- waited for at the end of this call only
            agreed = self._agree_start(labels)
        # (measure_exchange: two events on the compute stream bracket the waits for the exchanges -- the time this rank's
        # kernels are done and the Z pass cannot start yet = the EXPOSED part of the exchange; read with exposed_ms())
        timed = getattr(self, "measure_exchange", False) and labels.is_cuda
        if timed:
            self._ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self._ev[0].record()
        for req in pending:
            req.wait()
        if timed:
            self._ev[1].record()
        if use16:
            out = self._out16 if self.reuse_output else None
            if out is None or tuple(out.shape) != (self.sz, ye - ys, self.sx) or out.device != labels.device:
                out = torch.empty((self.sz, ye - ys, self.sx), dtype=torch.float32, device=labels.device)
                if self.reuse_output:
                    self._out16 = out
            self.ops.z_records16(dst, out, w, flags | (_lib.FLAG_SQRT if sqrt else 0))
            self._last_halo = halo
            if self._agree_finish(agreed) != 0:
                return None  # (some tile somewhere had no 16-bit form: the caller repeats the step with fp32 records)
            return out
        self._last_halo = halo
        self.ops.z_records(dst, self.sx, ye - ys, w[2], flags | (_lib.FLAG_SQRT if sqrt else 0), wxy=(w[0], w[1]))
        # the result is the float part of every record: a (sz, syl, sx) view with z-stride = record
        return dst[:, :(ye - ys) * self.sx].view(self.sz, ye - ys, self.sx)

    def _agree_start(self, labels):
        """MAX over the ranks of the refused-tile counter, then its copy to the host -- enqueued, not waited for."""
        flag = self._refused
        if self._stage and flag.is_cuda:   # (gloo: collectives on host tensors; a synchronising copy -- tests only)
            host = flag.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.MAX, group=self.group)
            self._refused_host.copy_(host)
            return None
        work = dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group, async_op=True)
        if not flag.is_cuda:
            work.wait()
            self._refused_host.copy_(flag)
            return None
        if getattr(self, "_agree_stream", None) is None:
            self._agree_stream = torch.cuda.Stream(labels.device)
        ev = torch.cuda.Event()
        with torch.cuda.stream(self._agree_stream):
            work.wait()                      # (this stream waits for the collective; the compute stream does not)
            self._refused_host.copy_(flag, non_blocking=True)
            ev.record()
        return ev

    def _agree_finish(self, ev):
        if ev is not None:
            ev.synchronize()
        return int(self._refused_host.item())

    def _records_chunk(self, k, labels, halo, w, flags, rec, y_splits, dst, pending, slot, use16=False):
        """XY phase of chunk k and the enqueue (not the wait) of its exchange, on the current stream."""
        zs = self.local_z()[0]
        c0, c1 = self._chunk(self.rank, k)
        blocks = []
        for h in range(self.world):
            if h == self.rank:
                blocks.append(dst[c0:c1])  # own part: straight into the receive buffer
                continue
            key = (k, h, use16)
            buf = self._send.get(key)
            if buf is None or buf.shape != (c1 - c0, rec[h]) or buf.device != labels.device:
                buf = self._send[key] = torch.empty((c1 - c0, rec[h]), dtype=dst.dtype, device=labels.device)
            blocks.append(buf)
        kw = {} if slot is None else {"slot": slot}
        if use16:
            self.ops.xy_records16(labels[c0 - zs:c1 - zs], halo, self.code, w, flags, y_splits, blocks, self._refused, **kw)
        else:
            self.ops.xy_records(labels[c0 - zs:c1 - zs], halo, self.code, w, flags, y_splits, blocks, **kw)
        recv = [dst[slice(*self._chunk(h, k))] for h in range(self.world)]
        if self._exchange == "alltoall":  # (also at world 1: a no-op that keeps the dry run honest)
            # one collective call per chunk (RCCL runs it as a group of sends / receives; every
            # peer pair has its own xGMI link); the own part is already in place -> empty entries
            # (the own entry is a one-element dummy rather than an empty tensor: every entry of the list is an
            # ordinary non-empty message, whatever the backend makes of zero-length ones)
            dummies = getattr(self, "_dummies", None)
            if dummies is None:
                dummies = self._dummies = {}
            dummy = dummies.get(dst.dtype)
            if dummy is None or dummy.device != dst.device:
                dummy = dummies[dst.dtype] = torch.zeros(2, dtype=dst.dtype, device=dst.device)
            ins = [dummy[0:1] if h == self.rank else blocks[h] for h in range(self.world)]
            outs = [dummy[1:2] if h == self.rank else recv[h] for h in range(self.world)]
            pending.append(dist.all_to_all(outs, ins, group=self.group, async_op=True))
        else:
            peers = [h for h in range(self.world) if h != self.rank]
            pending.append(self._p2p([(blocks[h], h) for h in peers], [(recv[h], h) for h in peers]))

    def exposed_ms(self):
        """Exposed exchange time of the last run() with ``measure_exchange`` set (after a device synchronisation):
        how long the compute stream sat between its last XY kernel and the arrival of the last record."""
        ev = getattr(self, "_ev", None)
        return float(ev[0].elapsed_time(ev[1])) if ev else None

    # -- the pipeline -----------------------------------------------------------------------
    def run(self, labels, weights_xyz, black_border=False, sqrt=False, gather_back=False):
        """Returns this rank's Y-slab of the result (all z, its rows) -- in the slab-record form a
        VIEW with a z-stride of one record, not a contiguous tensor -- or, with gather_back, its
        original Z-slab.

        Enqueue-only with fp32 records.  With records of 16-bit rows (`last_records16`) run() BLOCKS the host at its end
        until the ranks' agreement "every tile had a 16-bit form" has arrived (one 4-byte all-reduce behind the last
        exchange, read back on a side stream): the result may only be handed out once it is known to be complete, and a
        refused tile anywhere means the step is repeated here with fp32 records before run() returns.  The wait ends when
        the last exchange has landed, i.e. a Z phase before the step's kernels do; `fallbacks16` counts the repeats."""
        zs, ze = self.local_z()
        if tuple(labels.shape) != (ze - zs, self.sy, self.sx) or not labels.is_contiguous():
            raise ValueError(f"rank {self.rank}: expected a contiguous ({ze - zs}, {self.sy}, {self.sx}) slab")
        w = tuple(float(np.float32(v)) for v in weights_xyz)
        flags = (_lib.FLAG_BLACK_BORDER if black_border else 0)
        if self.records:
            use16 = self._use16(w)
            self.last_records16 = use16
            out = self._run_records(labels, w, flags, sqrt, None, None, use16, halo_start=lambda: self._halo_start(labels))
            halo = self._last_halo
            if out is None:
                # a tile without a 16-bit form somewhere: the same step with fp32 records (the halo is here already); the plan
                # stays on those for a while (see __init__: retry16)
                if self.retry16:
                    self._backoff16 = self._backoff16_next
                    self._backoff16_next = min(1024, 2 * self._backoff16_next)
                else:
                    self._allow16 = False
                self.last_records16 = False
                self.fallbacks16 += 1
                out = self._run_records(labels, w, flags, sqrt, halo, None, False)
            if gather_back:
                out = self._reshard([out], to_y=False)[0]
            return out
        halo = self._halo(labels)
        partial, zflags = self.ops.xy(labels, halo, self.code, w, flags)
        partial_y, zflags_y = self._reshard([partial, zflags], to_y=True)
        out = self.ops.z(partial_y, zflags_y, w[2], flags | (_lib.FLAG_SQRT if sqrt else 0), wxy=(w[0], w[1]))
        if gather_back:
            out = self._reshard([out], to_y=False)[0]
        return out


# ------------------------------------------------------------------------------------------
# bench.py's N > 1 leg
# ------------------------------------------------------------------------------------------
def global_extents(world: int, edge: int = 512):
    """One volume with edge^3 voxels per GPU (8 GPUs -> (2*edge)^3, BASELINE configs[3])."""
    ext = [edge, edge, edge]
    k, axis = world, 2
    while k > 1 and k % 2 == 0:
        ext[axis] *= 2
        axis = (axis - 1) % 3
        k //= 2
    ext[2] *= k  # odd remainder: stack along z
    return tuple(ext)
