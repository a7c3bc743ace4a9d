"""The package's pipelined loop (yolov3/pipeline.py: what bench.py times and what stream.detect_in_frames / the command
line run) against per-frame ``inference()`` -- the reference's one-frame-per-call path,
/root/reference/yolov3/inference.py:286-368 and __main__.py:159-165 -- and against the benchmark's own rate.
Need an MI355X: -m gpu."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

import yolov3
from yolov3 import weights as W
from yolov3.pipeline import Pipeline
from yolov3.synthdata import synth_frames

from golden_util import MODELS, ROOT

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
        if len(tickets) == in_flight:
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


def test_detect_in_frames_on_pinned_batches_is_the_benchmarked_loop():
    """the public loop on pinned frames runs at the rate bench.py reports (within 10 %) and returns what
    per-frame inference() returns.  yolov3 608 bf16, batches of 16: BASELINE.json configs[2]."""
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "10", "--no-cpu-baseline",
                           "--no-extras"], capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-2000:]
    bench = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][0])
    net = _net("yolov3", "bf16", obj_bias=-8.5)
    nb = 40
    pinned = [torch.from_numpy(synth_frames(200 + j, 16, 608, 608)).pin_memory() for j in range(6)]

    def batches(n):
        for j in range(n):
            yield pinned[j % len(pinned)]

    list(yolov3.detect_in_frames(net, batches(8)))                       # plans, buffers, first launches
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    results = list(yolov3.detect_in_frames(net, batches(nb)))
    rate = 16 * nb / (time.perf_counter() - t0)
    assert len(results) == 16 * nb
    assert rate > 0.9 * bench["value"], (rate, bench["value"])
    for j in (0, 5):
        for f in (0, 7, 15):
            want = yolov3.inference(net, pinned[j][f].numpy(), prob_thresh=0.05, nms_iou_thresh=0.3)[0]
            assert _same(results[16 * j + f], want), (j, f)
