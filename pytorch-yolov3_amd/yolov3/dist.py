"""Multi-GPU data parallelism for the detection path: one process per GPU, frames sharded
contiguously, ONE collective per batch -- an all-gather (RCCL over xGMI when the tensors live
on GPUs, gloo in the CPU tests) of fixed-size padded detection records.

The reference is single-device; frames are independent (BN uses running
statistics, NMS is per frame: /root/reference/yolov3/inference.py:346), so the only exchange
is the result gather.  Record = 8 x int32: x1, y1, x2, y2, float32 score bits, class,
prediction row, valid flag; ``kmax`` records per frame (payload = frames * kmax * 32 B per
rank: latency-bound, KBs), plus the true per-frame count so truncation is detectable.
"""
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


def all_gather_records(records, counts, world, group=None):
    """records (b, kmax, 8) int32 and counts (b,) int32 of THIS rank -> the same for all ranks,
    concatenated in rank order: (world*b, kmax, 8), (world*b,).  Every rank must pass equal b."""
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return records, counts          # plain single-process use: nothing to gather
    out_r = torch.empty((world * records.shape[0],) + tuple(records.shape[1:]), dtype=records.dtype,
                        device=records.device)
    out_c = torch.empty((world * counts.shape[0],), dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(out_r, records.contiguous(), group=group)
    dist.all_gather_into_tensor(out_c, counts.contiguous(), group=group)
    return out_r, out_c


def pack_records_host(dets, kmax):
    """Host-side packer with the layout of the device kernel ``y3_pack_records`` (used by the CPU
    tests of the collective plumbing; the product path packs on the GPU).
    dets: list of [tlbr (K,4), prob (K,), cls (K,), rows (K,)]."""
    b = len(dets)
    rec = np.zeros((b, kmax, RECORD_INTS), dtype=np.int32)
    cnt = np.zeros(b, dtype=np.int32)
    for i, d in enumerate(dets):
        k = min(len(d[1]), kmax)
        cnt[i] = len(d[1])
        rec[i, :k, 0:4] = d[0][:k]
        rec[i, :k, 4] = np.asarray(d[1][:k], dtype=np.float32).view(np.int32)
        rec[i, :k, 5] = d[2][:k]
        rec[i, :k, 6] = d[3][:k] if len(d) > 3 else 0
        rec[i, :k, 7] = 1
    return rec, cnt


def unpack_records(records, counts):
    """Inverse of the packers: -> per frame [tlbr int64 (K,4), prob f32 (K,), cls int64 (K,),
    rows int64 (K,), truncated bool]."""
    rec = records.cpu().numpy() if isinstance(records, torch.Tensor) else np.asarray(records)
    cnt = counts.cpu().numpy() if isinstance(counts, torch.Tensor) else np.asarray(counts)
    out = []
    for i in range(rec.shape[0]):
        k = int(min(cnt[i], rec.shape[1]))
        r = rec[i, :k]
        out.append([r[:, 0:4].astype(np.int64), np.ascontiguousarray(r[:, 4]).view(np.float32).copy(),
                    r[:, 5].astype(np.int64), r[:, 6].astype(np.int64), bool(cnt[i] > rec.shape[1])])
    return out


class DetectionGather(object):
    """Device buffers + the per-batch gather of one rank."""

    def __init__(self, batch, rows, kmax, device, world, group=None):
        self.batch, self.rows, self.kmax, self.world, self.group = batch, rows, kmax, world, group
        self.records = torch.zeros((batch, kmax, RECORD_INTS), dtype=torch.int32, device=device)
        self.counts = torch.zeros(batch, dtype=torch.int32, device=device)

    def run(self, det):
        """det: yolov3.inference.Detector after ``run``.  Returns (records, counts) of all ranks."""
        from . import _hip
        _hip.check(_hip.lib().y3_pack_records(
            det.count.data_ptr(), det.tlbr.data_ptr(), det.prob.data_ptr(), det.cls.data_ptr(), det.row.data_ptr(),
            self.batch, self.rows, self.kmax, self.records.data_ptr(), self.counts.data_ptr(), _hip.stream_ptr()))
        return all_gather_records(self.records, self.counts, self.world, self.group)
