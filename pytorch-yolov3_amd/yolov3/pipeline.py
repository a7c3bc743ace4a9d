"""``Pipeline``: several batches in flight between host frames and host detections.

What the reference does per frame in ``inference()`` (/root/reference/yolov3/inference.py:335 ``torch.tensor(inp, device=..)``,
:336 forward, :338-340 ``.detach().cpu().numpy()``, :342-366 threshold / NMS on the host) is here one pipelined pass per BATCH:

    pinned host frames --y3_copy_bytes (copy stream)--> device frame buffer --forward (stream k, arena k)--> y3_detect
        --> y3_pack_records [--all_gather_into_tensor on a side stream, N ranks--] --y3_copy_bytes--> pinned host records

with ``in_flight`` batches (default 3) on their own HIP streams, each with its own activation arena, detector buffers and
record buffers, so that one batch's kernel tails, its frame upload and its record download run under the other batches'
kernels.  This is the loop ``bench.py`` times (its headline number is this class fed from pinned host memory) and the loop
``stream.detect_in_frames`` / the command line run: the benchmarked configuration IS the package's.

Results equal per-frame ``inference()`` bit for bit (frames are independent, kernel choice does not change a bit:
tests/test_gpu_properties.py, tests/test_gpu_pipeline.py) -- with one rank or N: a frame that kept more than ``kmax`` boxes is
fetched again in full (one rank) or gathered again with records for the largest count (N ranks, ``results``: SURVEY.md 8(e)'s
"ragged alternative"; the counts ride in the first gather, so every rank takes the same decision).
"""
import warnings

import numpy as np
import torch

from . import _hip
from .dist import DetectionGather, all_gather_records, regather_if_truncated, unpack_records
from .inference import Detector


# HIP deals streams onto its hardware queues (GPU_MAX_HW_QUEUES, 8 here) in creation order and two streams on one queue run one
# after the other, so streams are a per-device resource: every Pipeline of a device uses the SAME compute / copy / gather
# streams (a second Pipeline with four streams of its own cost the first one 15 % -- and yolov3-tiny, whose kernels are short,
# 20 %: profiles/r04m_pipeline_depth.txt).  Pipelines that are active at the same time then share streams: still correct
# (everything is ordered by streams and events), just not concurrent with each other.
_STREAMS = {}


def _device_streams(dev, n_compute, want_side):
    pool = _STREAMS.setdefault((dev.type, dev.index), {"compute": [], "copy": None, "side": None})
    while len(pool["compute"]) < n_compute:
        pool["compute"].append(torch.cuda.Stream(device=dev))
    if pool["copy"] is None:
        pool["copy"] = torch.cuda.Stream(device=dev)
    if want_side and pool["side"] is None:
        pool["side"] = torch.cuda.Stream(device=dev)
    return pool["compute"][:n_compute], pool["copy"], pool["side"] if want_side else None


class Pipeline(object):
    """``submit(frames)`` enqueues one batch and returns a ticket; ``results(ticket)`` waits for that batch only.

    frames: a (batch, H, W, 3) uint8 BGR batch -- a PINNED torch tensor (uploaded straight from where it lies; fill
    ``host_frames(j)`` to get one), a device tensor (no upload), or anything else array-like (staged through a pinned buffer:
    one extra host copy).  ``in_flight`` batches run concurrently (streams, arenas); ``max_open`` = 2 x in_flight tickets may
    be open, each with its own detector and record buffers, so that every stream always has its NEXT batch queued behind the
    running one while the host is still reading an older batch's records (waiting for the oldest of only ``in_flight`` open
    tickets before every submit costs 8-15 % of the rate: profiles/r04m_pipeline_depth.txt).  Submitting ticket
    i + max_open reuses ticket i's buffers: its results must have been taken, or are dropped (the benchmark does that on
    purpose).
    """

    def __init__(self, net, batch, height=None, width=None, in_flight=3, prob_thresh=0.05, nms_iou_thresh=0.3, kmax=512,
                 world=1, group=None, options=None, copy_blocks=8):
        _hip.require_gpu()
        if batch < 1 or in_flight < 1:
            raise ValueError("batch and in_flight must be positive")
        if not str(net.device).startswith("cuda"):
            net.cuda()
        self.net, self.batch, self.in_flight, self.kmax = net, int(batch), int(in_flight), int(kmax)
        self.height = int(height or net.net_info["height"])
        self.width = int(width or net.net_info["width"])
        self.prob_thresh, self.nms_iou_thresh = float(np.float32(prob_thresh)), float(nms_iou_thresh)
        self.dev = dev = net._torch_device()
        self.group = group
        nq, sure = _hip.hw_queues()
        if in_flight > 1 and (nq < in_flight + 2 or not sure):
            warnings.warn("yolov3.Pipeline: {} batches in flight + a copy stream (+ a gather stream) want GPU_MAX_HW_QUEUES >= {}, "
                          "but the HIP runtime {} {} hardware queues (it reads the variable when it initialises: import yolov3 -- or "
                          "set GPU_MAX_HW_QUEUES=8 -- before the first torch.cuda call); streams that share a queue run one after "
                          "the other (measured: 4.9 k instead of 6.3 k frames/s)".format(
                              in_flight, in_flight + 2, "has" if sure else "was probably initialised with", nq), RuntimeWarning)
        self.copy_blocks = int(copy_blocks)
        # several batches in flight: CU time counts, not one launch's tail -> the strip kernel's 256-pixel tiles
        # (include/yolov3_hip.h: Y3_AM_HALO_TILE256; profiles/HISTORY.md 3.1c).  Explicit ``options`` win.
        if options is None and self.in_flight > 1:
            base = dict(net.options or {})
            base["auto_mask"] = int(base.get("auto_mask", _hip.options().auto_mask)) | _hip.AM_HALO_TILE256
            options = base
        self.options = dict(options) if options else None
        with torch.cuda.device(dev):
            distributed = world > 1 or (torch.distributed.is_available() and torch.distributed.is_initialized())
            # one side stream for all gathers of this rank: every HIP stream needs a hardware queue of its own to overlap
            self.streams, self.copy_stream, side = _device_streams(dev, self.in_flight, distributed)
            plans = [net._get_plan(self.batch, self.height, self.width, "u8", slot=k, options=self.options)
                     for k in range(self.in_flight)]
            self.rows = plans[0].rows_total
            self.max_open = 2 * self.in_flight
            self.dets = [Detector(self.batch, self.rows, dev) for _ in range(self.max_open)]
            self.gathers = [DetectionGather(self.batch, self.rows, self.kmax, dev, world, group=group, side=side)
                            for _ in range(self.max_open)]
            self.world = world
            # TWO device frame buffers per batch in flight: with one, the upload of ticket i could only start when the
            # forward of ticket i - in_flight (same stream, same buffer) had finished -- exactly when that stream was ready
            # for its next forward, which then waited out the whole copy (profiles/r03c_pcie_inclusive.txt)
            self.nbuf = 2 * self.in_flight
            shape = (self.batch, self.height, self.width, 3)
            self.dev_frames = [torch.empty(shape, dtype=torch.uint8, device=dev) for _ in range(self.nbuf)]
            self._host_frames = [None] * self.nbuf           # pinned staging / caller-fillable buffers, made on demand
            self.free_ev = [torch.cuda.Event() for _ in range(self.nbuf)]
            self.ready_ev = [torch.cuda.Event() for _ in range(self.nbuf)]
            # "has the upload that last READ this host buffer finished": one event per HOST buffer (keyed by its address), not per
            # device buffer -- the device-buffer rotation (uploads % nbuf) and a caller's own host-buffer rotation are unrelated
            # (a caller double-buffering with host_frames(0) / (1) got the event of a long-finished upload)
            self._host_ev = {}
            for e in self.free_ev + self.ready_ev:
                e.record()
            self.host_rec = [torch.empty((world * self.batch, self.kmax, 8), dtype=torch.int32).pin_memory()
                             for _ in range(self.max_open)]
            self.done_ev = [torch.cuda.Event() for _ in range(self.max_open)]
            self.full_hw = torch.tensor([[self.height, self.width]] * self.batch, dtype=torch.int32, device=dev)
            torch.cuda.synchronize(dev)
        self.busy = False               # set by a caller that owns the pipeline for a while (stream.detect_in_frames)
        self._n = 0                     # tickets issued
        self._uploads = 0               # uploads issued (device frame buffer = uploads % nbuf)
        self._frames_in = [0] * self.max_open
        self._on_host = [False] * self.max_open

    # ------------------------------------------------------------------ host buffers
    def host_frames(self, j):
        """Pinned (batch, H, W, 3) uint8 buffer number ``j`` (0 .. 2 * in_flight - 1) for the caller to decode frames into;
        ``submit`` uploads from it without a staging copy.  Refill buffer j only after ``upload_done(j)``."""
        j %= self.nbuf
        if self._host_frames[j] is None:
            self._host_frames[j] = torch.empty((self.batch, self.height, self.width, 3), dtype=torch.uint8).pin_memory()
        return self._host_frames[j]

    def upload_done(self, j, wait=True):
        """Has the last upload that read ``host_frames(j)`` finished (``wait``: block until it has)?"""
        buf = self._host_frames[j % self.nbuf]
        return True if buf is None else self.host_buffer_free(buf, wait)

    def host_buffer_free(self, host, wait=True):
        """Same question for ANY pinned tensor that was handed to ``submit``: may the caller overwrite it?"""
        ev = self._host_ev.get(host.data_ptr())
        if ev is None:
            return True
        if wait:
            ev.synchronize()
            return True
        return ev.query()

    def _record_host_read(self, host):
        ev = self._host_ev.get(host.data_ptr())
        if ev is None:
            if len(self._host_ev) >= 64:        # callers that pass a fresh pinned tensor every time: forget finished uploads
                for ptr in [p for p, e in self._host_ev.items() if e.query()]:
                    del self._host_ev[ptr]
            ev = self._host_ev[host.data_ptr()] = torch.cuda.Event()
        ev.record(self.copy_stream)

    def next_slot(self):
        """(stream, ticket buffer index) the NEXT ``submit`` will use: callers that prepare device frames themselves (resize on the
        GPU) do it on that stream and keep their tensor alive under that index until the buffers come round again."""
        return self.streams[self._n % self.in_flight], self._n % self.max_open

    # ------------------------------------------------------------------ the pipelined step
    def submit(self, frames, orig_hw=None, n_frames=None, to_host=True):
        """Enqueue one batch.  ``orig_hw``: (batch, 2) original frame sizes when the frames were resized to the network's
        size (default: they are net-sized).  ``n_frames``: how many leading frames of the batch are real (a short last
        batch).  ``to_host=False`` leaves the records on the device (resident-rate measurements).  Returns the ticket."""
        i = self._n
        k, d = i % self.in_flight, i % self.max_open        # stream / arena of the batch; detector / record buffers of the ticket
        lib = _hip.lib()
        with torch.cuda.device(self.dev), torch.cuda.stream(self.streams[k]):
            cur = self.streams[k]
            if isinstance(frames, torch.Tensor) and frames.is_cuda:
                fr, j = frames, None
            else:
                j = self._uploads % self.nbuf
                self._uploads += 1
                host = frames if (isinstance(frames, torch.Tensor) and frames.is_pinned()) else None
                if host is None:
                    # pageable input: stage it (the staging buffer is free once the upload that last READ IT has finished)
                    host = self.host_frames(j)
                    self.host_buffer_free(host, wait=True)
                    src = frames if isinstance(frames, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(frames))
                    if src.dim() != 4 or src.shape[0] > self.batch:
                        raise ValueError("expected at most {} frames of shape ({}, {}, 3), got {}".format(
                            self.batch, self.height, self.width, tuple(src.shape)))
                    host[:src.shape[0]].copy_(src)
                if tuple(host.shape[1:]) != (self.height, self.width, 3) or host.dtype != torch.uint8 or not host.is_contiguous():
                    raise ValueError("expected contiguous uint8 frames of shape (<= {}, {}, {}, 3)".format(self.batch, self.height, self.width))
                # the copy runs on its own stream, into the buffer the forward of upload (uploads - nbuf) read last
                with torch.cuda.stream(self.copy_stream):
                    self.copy_stream.wait_event(self.free_ev[j])
                    _hip.check(lib.y3_copy_bytes(host.data_ptr(), self.dev_frames[j].data_ptr(),
                                                 min(host.numel(), self.dev_frames[j].numel()), self.copy_blocks,
                                                 _hip.stream_ptr(self.copy_stream)))
                    self.ready_ev[j].record(self.copy_stream)
                    self._record_host_read(host)
                cur.wait_event(self.ready_ev[j])
                fr = self.dev_frames[j]
            if tuple(fr.shape) != (self.batch, self.height, self.width, 3):
                raise ValueError("device frames must be ({}, {}, {}, 3) uint8".format(self.batch, self.height, self.width))
            out = self.net.forward_frames(fr, fresh=False, slot=k, options=self.options)
            if j is not None:
                self.free_ev[j].record(cur)
            det = self.dets[d]
            det.run(out, self.full_hw if orig_hw is None else orig_hw, self.prob_thresh, self.nms_iou_thresh)
            if n_frames is not None and int(n_frames) < self.batch:
                # padding frames of a short last batch (copies of real frames, or whatever the staging buffer held) must not
                # count: their detections are dropped here, BEFORE the records are packed and gathered, so that they can neither
                # trigger the "some frame kept more than kmax boxes" second fetch / second collective nor reach another rank
                det.count[int(n_frames):].zero_()
            g = self.gathers[d]
            rec = g.run(det)
            if to_host and g.done is not None:
                # N ranks: the gathered records go to the host on the SIDE stream, right behind the all-gather -- the compute stream
                # never waits for the collective (until round 6 it did, and with it every batch queued behind on that stream).
                # (With ONE rank under a process group the all-gather itself costs 11 % either way: RCCL then runs it as a
                # device-to-device copy through the runtime's blit path, profiles/r06_gather_one_rank.txt.)
                with torch.cuda.stream(g.side):
                    _hip.check(lib.y3_copy_bytes(rec.data_ptr(), self.host_rec[d].data_ptr(), rec.numel() * rec.element_size(), 4,
                                                 _hip.stream_ptr(g.side)))
                    self.done_ev[d].record(g.side)
            else:
                if to_host:
                    _hip.check(lib.y3_copy_bytes(rec.data_ptr(), self.host_rec[d].data_ptr(), rec.numel() * rec.element_size(), 4,
                                                 _hip.stream_ptr(cur)))
                elif g.done is not None:
                    cur.wait_event(g.done)                # records left on the device: "done" still means gathered
                self.done_ev[d].record(cur)
        self._frames_in[d] = self.batch if n_frames is None else int(n_frames)
        self._on_host[d] = bool(to_host)
        self._n += 1
        return i

    def records(self, ticket):
        """Host records of all ranks for ``ticket``: (world * batch, kmax, 8) int32 (a view of the pinned buffer, valid
        until the ticket's slot is reused); waits for that batch only."""
        if not (self._n - self.max_open <= ticket < self._n):
            raise ValueError("ticket {} is not open (open: {} .. {})".format(ticket, max(0, self._n - self.max_open), self._n - 1))
        d = ticket % self.max_open
        if not self._on_host[d]:
            raise ValueError("ticket {} was submitted with to_host=False: its records stayed on the device".format(ticket))
        self.done_ev[d].synchronize()
        return self.host_rec[d].numpy()

    def results(self, ticket, return_rows=False):
        """``[bbox_tlbr int64 (K,4), class_prob f32 (K,), class_idx int64 (K,)]`` (+ prediction rows) per frame, the contract of
        ``inference()``.  One rank: the ``n_frames`` real frames of this batch.  Under a process group (``world`` > 1): the
        frames of ALL ranks in rank order, ``world * batch`` lists (padding frames of a short batch come back empty).  A frame
        that kept more than ``kmax`` boxes is fetched again in full -- under a process group that second fetch is a second
        all-gather, i.e. a COLLECTIVE: every rank must call ``results()`` (not just ``records()``) for the same tickets in the
        same order, or the job hangs.  All ranks take the same branch because the first gather carried every frame's true count."""
        rec = self.records(ticket)
        k, d = ticket % self.in_flight, ticket % self.max_open
        mine = rec[:self.batch] if self.world == 1 else rec     # single rank: all frames are ours
        collective = self.gathers[d].collective                # a process group is up (N ranks, or one: tests)
        if not collective and mine.size and int(mine[:, 0, 7].max()) > self.kmax:
            # (the ticket's detector buffers still hold the whole batch: they are reused max_open tickets later)
            with torch.cuda.device(self.dev), torch.cuda.stream(self.streams[k]):
                full = self.dets[d].fetch(return_rows=return_rows, kmax=int(mine[:, 0, 7].max()))
            return full[:self._frames_in[d]]
        if collective:
            # N ranks: every kept box of every rank's frames (the reference returns them all: inference.py:355-366).  The true
            # counts came with the first gather, so all ranks see the same maximum and take the same branch: when a frame kept
            # more than kmax boxes, every rank packs its batch again with room for the largest count and the records are
            # gathered a second time (a collective: all ranks call results() for the same tickets in the same order).
            det = self.dets[d]

            def repack(kmax2):
                with torch.cuda.device(self.dev), torch.cuda.stream(self.streams[k]):
                    rec2 = torch.zeros((self.batch, kmax2, 8), dtype=torch.int32, device=self.dev)
                    _hip.check(_hip.lib().y3_pack_records(
                        det.count.data_ptr(), det.tlbr.data_ptr(), det.prob.data_ptr(), det.cls.data_ptr(), det.row.data_ptr(),
                        self.batch, self.rows, kmax2, rec2.data_ptr(), None, _hip.stream_ptr(self.streams[k])))
                return rec2
            with torch.cuda.device(self.dev), torch.cuda.stream(self.streams[k]):
                mine = regather_if_truncated(mine, self.kmax, repack, self.world, self.group)
        items = unpack_records(mine)
        if any(item[4] for item in items):
            raise RuntimeError("Pipeline.results: a frame was truncated to kmax = {} boxes".format(self.kmax))
        n = self._frames_in[d] if self.world == 1 else len(items)
        return [item[:4] if return_rows else item[:3] for item in items[:n]]

    def synchronize(self):
        for s in self.streams + [self.copy_stream]:
            s.synchronize()
        side = self.gathers[0].side if self.gathers else None
        if side is not None:                              # (N ranks: the gathers and the records' way home run there)
            side.synchronize()
