"""Shared helpers for tests that read tests/golden/* (see tools/make_goldens.py)."""
import hashlib
import os
from collections import Counter

import numpy as np

from yolov3 import weights as W
from yolov3.cfgparse import parse_config

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLDEN = os.path.join(ROOT, "tests", "golden")
MODEL_DIR = os.path.join(ROOT, "pytorch-yolov3_amd", "models")
MODELS = {m: os.path.join(MODEL_DIR, m + ".cfg") for m in ("yolov3-tiny", "yolov3", "yolov3-spp")}
MODELS["mini"] = os.path.join(GOLDEN, "cfg", "mini.cfg")
MODEL_DIMS = {"yolov3-tiny": 416, "yolov3": 608, "yolov3-spp": 608}
GOLDEN_SEED = 0
GOLDEN_OBJ_BIAS = -5.0


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def load_jpeg_bgr(name):
    from PIL import Image
    im = Image.open(os.path.join(GOLDEN, "images", name)).convert("RGB")
    return np.ascontiguousarray(np.asarray(im)[:, :, ::-1])


# objectness bias of the "bench regime" goldens (tools/make_goldens.py: BENCH_OBJ_BIAS): a few hundred candidates per frame
BENCH_REGIME_OBJ_BIAS = {"yolov3": -8.5, "yolov3-spp": -8.5, "yolov3-tiny": -5.0}
SAMPLE_IMAGES = ["000000035279.jpg", "000000078170.jpg", "000000229358.jpg", "000000253835.jpg", "000000377368.jpg",
                 "000000393569.jpg", "000000410880.jpg", "000000529762.jpg", "000000547336.jpg"]


def golden_params(model, obj_bias=GOLDEN_OBJ_BIAS):
    """The procedural parameters the golden vectors were generated with."""
    blocks, net_info = parse_config(MODELS[model])
    calib = None if model == "mini" else W.load_calibration(model)
    return W.synth_params(blocks, net_info, seed=GOLDEN_SEED, obj_bias=obj_bias, calib=calib)


def golden_weights_path(model, tmp_path=None, obj_bias=GOLDEN_OBJ_BIAS):
    """Write (once per machine) the Darknet-format file the golden run loaded, so tests exercise
    the ordinary ``load_weights`` path."""
    cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), "y3_golden_weights")
    os.makedirs(cache, exist_ok=True)
    tag = "" if obj_bias == GOLDEN_OBJ_BIAS else "_ob%g" % obj_bias
    path = os.path.join(cache, "%s_seed%d%s.weights" % (model, GOLDEN_SEED, tag))
    blocks, net_info = parse_config(MODELS[model])
    want = 20 + 4 * W.stream_length(blocks, net_info)
    if not (os.path.exists(path) and os.path.getsize(path) == want):
        tmp = path + ".%d.tmp" % os.getpid()
        W.write_darknet_weights(tmp, golden_params(model, obj_bias))
        os.replace(tmp, path)
    return path


def bench_regime_frame(name, dim):
    """Frame of a bench-regime golden entry: one of the nine sample_dataset JPEGs (original size) or a procedural frame."""
    from yolov3.synthdata import synth_frames
    if name.startswith("img"):
        return load_jpeg_bgr("000000%s.jpg" % name[3:])
    if name.startswith("crop"):                     # net-sized centre crop (tools/make_goldens.py: center_crop): never resized
        frame = load_jpeg_bgr("000000%s.jpg" % name[4:])
        y0, x0 = (frame.shape[0] - dim) // 2, (frame.shape[1] - dim) // 2
        return np.ascontiguousarray(frame[y0:y0 + dim, x0:x0 + dim])
    return synth_frames(int(name[5:]), 1, dim, dim)[0]


def product_candidates(bbox, prob, cls, orig_shape, prob_thresh):
    """Candidates of one frame as the reference's ``inference()`` forms them (inference.py:342-353, 269-283) from
    forward outputs: rows with score >= threshold, their truncated pixel boxes and classes."""
    mask = prob >= np.float32(prob_thresh)
    rows = np.where(mask)[0]
    box = bbox[mask].astype(np.float32).copy()
    box[:, [0, 2]] *= orig_shape[1]
    box[:, [1, 3]] *= orig_shape[0]
    xywh = box.astype(np.int64)
    half = xywh[:, 2:4] // 2
    return rows, np.concatenate([xywh[:, 0:2] - half, xywh[:, 0:2] + half], axis=1), cls[mask]


def _overlap(a, b):
    return min(a[2], b[2]) - max(a[0], b[0]) + 1 > 0 and min(a[3], b[3]) - max(a[1], b[1]) + 1 > 0


def compare_detections(g, prefix, det, rows=None, prob_tol=1e-5, cand=None):
    """One frame's detections against the reference's (G7 goldens).

    Integer work is exact: the same prediction rows, classes and pixel boxes.  The one legitimate source of a
    difference is upstream: the float32 forward differs from the reference's by summation order (gated at 1e-3,
    measured < 1e-4), and where a candidate's scaled coordinate (or its score at the threshold) lies that close
    to an integer, ``astype(int)`` lands on the other side -- a FLIP, visible by comparing every candidate's
    truncated box with the reference's (``cand`` = ``product_candidates`` of the product's own forward outputs;
    the goldens store the reference's).  Rule: with no flip the keep sets are identical; otherwise every
    differing row is itself a flip or hangs on one through a chain of same-class overlaps among the differing
    rows, and there are at most 4 of them per flip.  Without ``cand`` (the oracle on the golden machine): exact.
    """
    tlbr, prob, cls = det[0], det[1], det[2]
    g_tlbr, g_prob, g_cls, g_rows = g[prefix + "tlbr"], g[prefix + "prob"], g[prefix + "cls"], g[prefix + "rows"]
    assert tlbr.dtype == np.int64 and cls.dtype == np.int64 and prob.dtype == np.float32
    assert tlbr.shape == (len(prob), 4) and cls.shape == prob.shape
    if rows is None:
        got = Counter((int(c),) + tuple(int(v) for v in t) for c, t in zip(cls, tlbr))
        want = Counter((int(c),) + tuple(int(v) for v in t) for c, t in zip(g_cls, g_tlbr))
        diff = sum(((got - want) + (want - got)).values())
        assert diff == 0, "detections differ in %d rows" % diff
        return 0, 0
    got = {int(r): k for k, r in enumerate(rows)}
    want = {int(r): k for k, r in enumerate(g_rows)}
    assert len(got) == len(rows), "duplicate rows in detections"
    ref_box = {int(r): (tuple(int(v) for v in t), int(c)) for r, t, c in
               zip(g[prefix + "cand_rows"], g[prefix + "cand_tlbr"], g[prefix + "cand_cls"])}
    flipped = set()
    my_box = dict(ref_box)
    if cand is not None:
        my_box = {int(r): (tuple(int(v) for v in t), int(c)) for r, t, c in zip(*cand)}
        flipped = {r for r in set(my_box) | set(ref_box) if my_box.get(r) != ref_box.get(r)}
    diff = set(got) ^ set(want)
    if diff:
        assert flipped, "keep sets differ at rows %s although every candidate box equals the reference's" % sorted(diff)[:8]
        assert len(diff) <= 4 * len(flipped), "%d rows differ for %d flipped candidates" % (len(diff), len(flipped))
        nodes = {r: (my_box.get(r) or ref_box[r]) for r in diff | flipped}
        reached, frontier = set(flipped), list(flipped)
        while frontier:
            r = frontier.pop()
            for q in nodes:
                if q not in reached and nodes[q][1] == nodes[r][1] and _overlap(nodes[q][0], nodes[r][0]):
                    reached.add(q)
                    frontier.append(q)
        assert diff <= reached, "rows %s differ without a flipped candidate to explain them" % sorted(diff - reached)[:8]
    bad = 0
    for r in set(got) & set(want):
        a, b = got[r], want[r]
        assert cls[a] == g_cls[b], "class differs at row %d" % r
        assert abs(float(prob[a]) - float(g_prob[b])) <= prob_tol, "score differs at row %d" % r
        if not (tlbr[a] == g_tlbr[b]).all():
            assert r in flipped and np.abs(tlbr[a] - g_tlbr[b]).max() <= 1, \
                "box differs at row %d: %s vs %s" % (r, tlbr[a], g_tlbr[b])
            bad += 1
    return len(diff), bad, len(flipped)


def bf16_agreement(emulate="bf16"):
    """tests/golden/bf16_agreement.json / f16_agreement.json (tools/make_bf16_fixture.py [--emulate f16]): what the oracle
    emulating that storage type reaches against the reference's float32 detections on the G7 frames -- the floor for the
    HIP bf16 / fp16 path."""
    import json
    with open(os.path.join(GOLDEN, "%s_agreement.json" % emulate)) as fh:
        return json.load(fh)
