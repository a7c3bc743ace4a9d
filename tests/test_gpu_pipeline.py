"""The package's pipelined loop (yolov3/pipeline.py: what bench.py times and what stream.detect_in_frames / the command
line run) against per-frame ``inference()`` -- the reference's one-frame-per-call path,
/root/reference/yolov3/inference.py:286-368 and __main__.py:159-165 -- the tests that start child processes are in test_gpu_a_fresh_process.py.
Need an MI355X: -m gpu."""
import os

import numpy as np
import pytest
import torch

import yolov3
from yolov3 import weights as W
from yolov3.pipeline import Pipeline
from yolov3.synthdata import synth_frames

from golden_util import MODELS

pytestmark = pytest.mark.gpu


def _net(model, dtype, obj_bias=-5.0):
    net = yolov3.Darknet(MODELS[model], device="cuda", dtype=dtype).eval()
    net.set_params(W.synth_params(net.blocks, net.net_info, seed=0, obj_bias=obj_bias, calib=W.load_calibration(model)))
    return net


def _same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b)) and len(a) == len(b)


@pytest.mark.parametrize("in_flight", [1, 3])
def test_pipeline_equals_per_frame_inference(in_flight):
    """Pinned batches, pageable batches and device batches through the pipeline (several tickets open, slots reused):
    every frame's detections are bit-equal to ``inference()`` on that frame alone."""
    net = _net("yolov3-tiny", "float32")
    frames = synth_frames(31, 20, 416, 416)
    want = [yolov3.inference(net, f, prob_thresh=0.1, nms_iou_thresh=0.3, return_rows=True)[0] for f in frames]
    pipe = Pipeline(net, 4, in_flight=in_flight, prob_thresh=0.1, nms_iou_thresh=0.3)
    tickets = []
    got = []
    for b in range(5):
        chunk = frames[4 * b:4 * b + 4]
        if b % 3 == 0:
            src = pipe.host_frames(b)
            pipe.upload_done(b)
            src.copy_(torch.from_numpy(chunk))
        elif b % 3 == 1:
            src = chunk                                   # pageable numpy: staged by the pipeline
        else:
            src = torch.from_numpy(chunk).cuda()          # already on the device
        if len(tickets) == pipe.max_open:
            got += pipe.results(tickets.pop(0), return_rows=True)
        tickets.append(pipe.submit(src))
    for t in tickets:
        got += pipe.results(t, return_rows=True)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert _same(g, w)


def test_pipeline_refetches_frames_with_more_than_kmax_detections():
    net = _net("yolov3-tiny", "float32", obj_bias=-2.0)
    frames = synth_frames(5, 2, 416, 416)
    want = [yolov3.inference(net, f, prob_thresh=0.05, nms_iou_thresh=0.3)[0] for f in frames]
    assert max(len(w[1]) for w in want) > 8
    pipe = Pipeline(net, 2, in_flight=2, prob_thresh=0.05, nms_iou_thresh=0.3, kmax=8)
    got = pipe.results(pipe.submit(frames))
    for g, w in zip(got, want):
        assert _same(g, w)


def test_padding_frames_of_a_short_batch_do_not_count():
    """A short last batch (n_frames < batch) is padded up to the pipeline's batch; whatever the padding frames hold -- here
    dense frames that keep far more than kmax boxes -- their detections are dropped before the records are packed: the records
    carry a count of zero for them, they cannot trigger the "a frame kept more than kmax boxes" second fetch (under a process
    group: a second all-gather on every rank), and ``results`` returns exactly the real frames."""
    net = _net("yolov3-tiny", "float32", obj_bias=-2.0)
    frames = synth_frames(17, 4, 416, 416)
    pipe = Pipeline(net, 4, in_flight=2, kmax=2048)
    full = pipe.records(pipe.submit(frames)).copy()
    assert int(full[:, 0, 7].min()) > 8                           # every frame keeps more boxes than the small kmax below
    small = Pipeline(net, 4, in_flight=2, kmax=8)                 # kmax far below what a frame keeps
    t = small.submit(frames, n_frames=1)
    rec = small.records(t)
    assert int(rec[0, 0, 7]) == int(full[0, 0, 7]) and (rec[1:, :, 7] == 0).all(), rec[:, 0, 7]
    res = small.results(t)                                        # the real frame IS over kmax: fetched again in full
    assert len(res) == 1 and len(res[0][1]) == int(full[0, 0, 7])
    t = small.submit(frames[:1].repeat(4, axis=0), n_frames=1)    # the same through a fresh ticket
    assert len(small.results(t)) == 1


def test_upload_done_follows_the_host_buffer_not_the_device_buffer():
    """``upload_done(j)`` must answer for the upload that last READ ``host_frames(j)``.  A caller that
    double-buffers with host buffers 0 / 1 while the pipeline rotates six device buffers: with the copy stream held up (a long
    sleep kernel queued on it), the third upload -- host buffer 0 again, device buffer 2 -- is still pending, so buffer 0 must
    NOT be reported free (it was, from the long-finished first upload's event), and the results stay those of the frames."""
    net = _net("yolov3-tiny", "float32")
    frames = synth_frames(77, 16, 416, 416)
    want = [yolov3.inference(net, f, prob_thresh=0.1, nms_iou_thresh=0.3, return_rows=True)[0] for f in frames]
    pipe = Pipeline(net, 2, in_flight=3, prob_thresh=0.1, nms_iou_thresh=0.3)
    tickets, got = [], []
    for b in range(8):
        j = b % 2
        host = pipe.host_frames(j)
        assert pipe.upload_done(j)                               # waits for the upload that read THIS buffer
        host.copy_(torch.from_numpy(frames[2 * b:2 * b + 2]))
        if b == 2:
            with torch.cuda.stream(pipe.copy_stream):
                torch.cuda._sleep(200_000_000)                   # ~0.1 s: the next upload queues behind it
        if len(tickets) == pipe.max_open:
            got += pipe.results(tickets.pop(0), return_rows=True)
        tickets.append(pipe.submit(host))
        if b == 2:
            assert not pipe.upload_done(0, wait=False), "host buffer 0 reported free while its upload is still queued"
            assert pipe.upload_done(1, wait=False)               # buffer 1's upload (b = 1) finished long ago
    for t in tickets:
        got += pipe.results(t, return_rows=True)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert _same(g, w)


def test_plans_compiled_back_to_back_keep_their_zero_page():
    """Regression (round 4): every plan holds the address of the network's 4-KiB zero page (padding taps, tile tails).
    Compiling a second plan used to allocate a new page and free the old one under the first plan; once other small
    tensors landed in that memory the first plan's border pixels were wrong -- silently, and only in plans compiled BEFORE
    another one (the pipeline compiles three in a row).  Outputs of several plans of one network must agree bit for bit
    whatever was allocated in between."""
    net = _net("yolov3-tiny", "bf16")
    frames = torch.from_numpy(synth_frames(77, 4, 416, 416)).cuda()
    plans = [net._get_plan(4, 416, 416, "u8", slot=k) for k in range(4)]         # compiled before anything runs
    junk = [torch.full((1024,), 255, dtype=torch.uint8, device="cuda") for _ in range(64)]   # what would land in freed pages
    outs = []
    for k in range(4):
        o = net.forward_frames(frames, fresh=True, slot=k)
        outs.append(o)
    torch.cuda.synchronize()
    assert len({p.handle.value for p in plans}) == 4 and junk[0][0] == 255
    for k in range(1, 4):
        for key in ("bbox_xywh", "class_prob", "class_idx"):
            assert torch.equal(outs[k][key], outs[0][key]), (k, key)


def test_copy_bytes_rejects_pageable_host_memory():
    """a pageable host pointer would be a GPU memory fault inside the copy kernel; y3_copy_bytes checks what the
    runtime knows about both ends and returns Y3_ERR_INVALID instead."""
    from yolov3 import _hip
    lib = _hip.lib()
    dst = torch.empty(4096, dtype=torch.uint8, device="cuda")
    pageable = torch.zeros(4096, dtype=torch.uint8)
    pinned = torch.arange(4096, dtype=torch.int32).to(torch.uint8).pin_memory()
    assert lib.y3_copy_bytes(pageable.data_ptr(), dst.data_ptr(), 4096, 1, _hip.stream_ptr()) != 0
    assert b"pinned" in lib.y3_last_error()
    _hip.check(lib.y3_copy_bytes(pinned.data_ptr(), dst.data_ptr(), 4096, 1, _hip.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(dst.cpu(), pinned)


def test_pipeline_is_deterministic_over_many_submits():
    """Soak: 72 batches through three streams / six tickets, three different inputs in rotation, the host reading results as it
    goes -- every occurrence of an input must give byte-identical records (a race between upload, forward, detection tail and
    record download, or a plan reading memory it does not own, shows up as a difference between occurrences)."""
    net = _net("yolov3-tiny", "bf16")
    pipe = Pipeline(net, 8, in_flight=3, prob_thresh=0.05, nms_iou_thresh=0.3)
    inputs = []
    for j in range(3):
        buf = pipe.host_frames(j)
        buf.copy_(torch.from_numpy(synth_frames(500 + j, 8, 416, 416)))
        inputs.append(buf)
    first = {}
    tickets = []

    def take(t, j):
        rec = pipe.records(t).copy()
        if j in first:
            assert np.array_equal(rec, first[j]), (t, j)
        else:
            first[j] = rec
            assert int(rec[:, 0, 7].max()) > 0          # there are detections to compare
    for i in range(72):
        if len(tickets) == pipe.max_open:
            take(*tickets.pop(0))
        tickets.append((pipe.submit(inputs[i % 3]), i % 3))
    for t, j in tickets:
        take(t, j)
    assert len(first) == 3 and not np.array_equal(first[0], first[1])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the single-GPU boxes skip it; an 8-GPU node runs it)")
def test_network_on_a_device_that_is_not_current():
    """plan compilation allocates and launches (arena, zero page, fragment-order weights) and must do so on
    the NETWORK's device, whichever device is current: a net on cuda:1 compiled and run while cuda:0 is current gives the bits
    of the same net on cuda:0, and nothing of it lives on GPU 0."""
    import yolov3
    from yolov3 import weights as W
    from yolov3.synthdata import synth_frames
    cfg = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch-yolov3_amd", "models", "yolov3.cfg")
    frames = synth_frames(5, 16, 608, 608)
    outs = []
    torch.cuda.set_device(0)
    for dev in ("cuda:0", "cuda:1"):
        net = yolov3.Darknet(cfg, device=dev, dtype="bf16").eval()
        net.set_params(W.synth_params(net.blocks, net.net_info, seed=0, obj_bias=-5.0, calib=W.load_calibration("yolov3")))
        assert torch.cuda.current_device() == 0
        out = net.forward_frames(torch.from_numpy(frames).to(dev))
        assert any("conv_halo_dw" in r["kernel"] or "conv1x1_dw" in r["kernel"] for r in net.plan_report())
        assert all(v.device == torch.device(dev) for v in out.values())
        for key, w in net._dev_weights.items():
            t = w if isinstance(w, torch.Tensor) else w["weight"]
            assert t.device == torch.device(dev), key
        outs.append({k: v.cpu() for k, v in out.items()})
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
