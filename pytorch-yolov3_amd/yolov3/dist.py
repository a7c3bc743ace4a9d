"""Multi-GPU data parallelism for the detection path: one process per GPU, frames sharded
contiguously, ONE collective per batch -- an all-gather (RCCL over xGMI when the tensors live
on GPUs, gloo in the CPU tests) of fixed-size padded detection records.

The reference is single-device; frames are independent (BN uses running
statistics, NMS is per frame: /root/reference/yolov3/inference.py:346), so the only exchange
is the result gather.  Record = 8 x int32: x1, y1, x2, y2, float32 score bits, class,
prediction row, and the frame's TRUE detection count (0 in padding records, so it doubles as
the valid flag, and a count above ``kmax`` says the frame was truncated); ``kmax`` records
per frame (payload = frames * kmax * 32 B per rank: latency-bound, KBs).  The count rides in
the records, so a batch costs exactly one ``all_gather_into_tensor``; on GPUs it is issued on
a side stream behind an event, so the compute stream goes straight on to the next batch.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

RECORD_INTS = 8


def shard_range(n_frames, rank, world):
    """Contiguous split of ``n_frames`` over ``world`` ranks: rank r owns [lo, hi)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world {}/{}".format(rank, world))
    base, extra = divmod(n_frames, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def counts_of(records):
    """Per-frame true detection counts carried in field 7 of each frame's first record."""
    return records[:, 0, 7]


def all_gather_records(records, world, group=None):
    """records (b, kmax, 8) int32 of THIS rank -> the records of all ranks concatenated in rank order,
    (world*b, kmax, 8), with ONE collective.  Every rank must pass equal b."""
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return records                  # plain single-process use: nothing to gather
    out = torch.empty((world * records.shape[0],) + tuple(records.shape[1:]), dtype=records.dtype,
                      device=records.device)
    dist.all_gather_into_tensor(out, records.contiguous(), group=group)
    return out


def regather_if_truncated(gathered, kmax, repack, world, group=None, round_to=64):
    """The "ragged alternative" of SURVEY.md 8(e), on top of the fixed-size gather: ``gathered`` (world*b, kmax, 8) are the
    records of all ranks; field 7 carries every frame's TRUE count, so every rank reads the same maximum.  If some frame kept
    more than ``kmax`` boxes, every rank calls ``repack(kmax2)`` -> its own (b, kmax2, 8) records with room for the largest
    count (rounded up to ``round_to``) and ONE more all-gather returns all of them; otherwise ``gathered`` comes back as it
    is.  Collective: all ranks call it, in the same order.  Nothing is ever dropped (the reference returns every kept box:
    /root/reference/yolov3/inference.py:355-366).  Returns host or device records like ``repack`` / ``gathered``."""
    cnt = counts_of(gathered)
    most = int(cnt.max()) if cnt.shape[0] else 0
    if most <= kmax:
        return gathered
    kmax2 = (most + round_to - 1) // round_to * round_to
    rec2 = repack(kmax2)
    if not isinstance(rec2, torch.Tensor):
        rec2 = torch.from_numpy(np.ascontiguousarray(rec2))
    out = all_gather_records(rec2, world, group)
    return out.cpu().numpy() if out.is_cuda else out


def pack_records_host(dets, kmax):
    """Host-side packer with the layout of the device kernel ``y3_pack_records`` (used by the CPU
    tests of the collective plumbing; the product path packs on the GPU).
    dets: list of [tlbr (K,4), prob (K,), cls (K,), rows (K,)]."""
    b = len(dets)
    rec = np.zeros((b, kmax, RECORD_INTS), dtype=np.int32)
    for i, d in enumerate(dets):
        k = min(len(d[1]), kmax)
        rec[i, :k, 0:4] = d[0][:k]
        rec[i, :k, 4] = np.asarray(d[1][:k], dtype=np.float32).view(np.int32)
        rec[i, :k, 5] = d[2][:k]
        rec[i, :k, 6] = d[3][:k] if len(d) > 3 else 0
        rec[i, :k, 7] = len(d[1])
    return rec


def unpack_records(records):
    """Inverse of the packers: -> per frame [tlbr int64 (K,4), prob f32 (K,), cls int64 (K,),
    rows int64 (K,), truncated bool]."""
    rec = records.cpu().numpy() if isinstance(records, torch.Tensor) else np.asarray(records)
    cnt = rec[:, 0, 7]
    out = []
    for i in range(rec.shape[0]):
        k = int(min(cnt[i], rec.shape[1]))
        r = rec[i, :k]
        out.append([r[:, 0:4].astype(np.int64), np.ascontiguousarray(r[:, 4]).view(np.float32).copy(),
                    r[:, 5].astype(np.int64), r[:, 6].astype(np.int64), bool(cnt[i] > rec.shape[1])])
    return out


class DetectionGather(object):
    """Device buffers + the per-batch gather of one rank.

    ``run`` packs on the CURRENT (compute) stream of the buffers' device, then issues the all-gather on a side stream
    behind an event, and returns the gathered tensor; ``done`` is recorded on the side stream when the gather has
    finished: consumers on other streams ``wait_event(done)`` (or synchronise the device) before reading it.
    Re-running the same object first waits for its previous gather, so the record buffer is never overwritten while
    RCCL still reads it -- and for ``consumed`` (``mark_consumed()``), if the caller read the previous result on some
    other stream, so that result is never overwritten under a reader either.
    ``side``: a stream shared by several gathers of one rank (HIP maps streams onto a handful of hardware queues;
    one side stream per batch in flight would make compute streams share queues and serialise them)."""

    def __init__(self, batch, rows, kmax, device, world, group=None, side=None):
        self.batch, self.rows, self.kmax, self.world, self.group = batch, rows, kmax, world, group
        self.device = torch.device(device)
        self.records = torch.zeros((batch, kmax, RECORD_INTS), dtype=torch.int32, device=device)
        self.gathered = self.records
        self.collective = world > 1 or (dist.is_available() and dist.is_initialized())
        self.side = self.packed = self.done = self.consumed = None
        if self.collective and self.device.type == "cuda":
            self.gathered = torch.empty((world * batch, kmax, RECORD_INTS), dtype=torch.int32, device=device)
            self.side = side if side is not None else torch.cuda.Stream(device=device)
            self.packed = torch.cuda.Event()
            self.done = torch.cuda.Event()
            self.done.record(self.side)

    def mark_consumed(self, stream=None):
        """Call after enqueueing the last read of the gathered tensor on a stream OTHER than the one ``run`` is called
        on (reads on that stream are ordered anyway): the next ``run`` waits for it before overwriting the result."""
        if self.device.type == "cuda":
            self.consumed = torch.cuda.Event()
            self.consumed.record(stream if stream is not None else torch.cuda.current_stream(self.device))

    def run(self, det):
        """det: yolov3.inference.Detector after ``run``.  Returns the records of all ranks, (world*batch, kmax, 8)."""
        from . import _hip
        cur = torch.cuda.current_stream(self.device) if self.device.type == "cuda" else None
        if self.side is not None:
            cur.wait_event(self.done)           # the previous gather of this buffer has been read by RCCL
        if self.consumed is not None:
            cur.wait_event(self.consumed)       # ... and its result by whoever took it to another stream
            self.consumed = None
        _hip.check(_hip.lib().y3_pack_records(
            det.count.data_ptr(), det.tlbr.data_ptr(), det.prob.data_ptr(), det.cls.data_ptr(), det.row.data_ptr(),
            self.batch, self.rows, self.kmax, self.records.data_ptr(), None, _hip.stream_ptr(cur)))
        if not self.collective:
            return self.records
        if self.side is None:
            return all_gather_records(self.records, self.world, self.group)
        self.packed.record(cur)
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.packed)
            dist.all_gather_into_tensor(self.gathered, self.records, group=self.group)
            self.done.record(self.side)
        return self.gathered
