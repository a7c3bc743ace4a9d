"""Pins the CPU oracle (oracle/darknet_oracle.py) against golden vectors produced by the REAL
reference in the build container (tools/make_goldens.py) and against the reference's own
known-answer test.  CPU only (-m "not gpu")."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import darknet_oracle as orc
from yolov3 import weights as W
from yolov3.preprocess import resize_bilinear_u8
from yolov3.synthdata import synth_frames

from golden_util import (GOLDEN, MODELS, MODEL_DIMS, golden_weights_path, load_jpeg_bgr, sha,
                         compare_detections)


def test_cxywh_to_tlbr_reference_known_answer():
    # /root/reference/tests/test_inference.py:12-23
    xywh = np.array([[5, 8, 10, 13, 10000], [100, 200, 30, 17, 19000]], dtype=np.int64)
    want = np.array([[0, 2, 10, 14, 10000], [85, 192, 115, 208, 19000]])
    assert (orc.cxywh_to_tlbr(xywh) == want).all()


def test_nms_cases_match_reference():
    with open(os.path.join(GOLDEN, "nms_cases.json")) as fh:
        cases = json.load(fh)
    assert len(cases) >= 10
    for c in cases:
        boxes = np.array(c["boxes"], dtype=np.int64).reshape(-1, 4)
        prob = np.array(c["prob_bits"], dtype=np.uint32).view(np.float32)
        got = [int(i) for i in orc.non_max_suppression(boxes, prob, iou_thresh=c["thr"])]
        assert got == c["agnostic"], c["name"]
        if c["cls"] is not None:
            cls = np.array(c["cls"], dtype=np.int64)
            got = [int(i) for i in orc.non_max_suppression(boxes, prob, class_idx=cls, iou_thresh=c["thr"])]
            assert sorted(got) == sorted(c["per_class"]), c["name"]


def test_nms_empty_and_survey_example():
    assert orc.non_max_suppression(np.zeros((0, 4), dtype=np.int64), np.zeros(0, dtype=np.float32),
                                   class_idx=np.zeros(0, dtype=np.int64)) == []
    boxes = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10]])
    prob = np.array([.9, .8, .7, .9], dtype=np.float32)
    assert sorted(orc.non_max_suppression(boxes, prob, class_idx=np.array([1, 1, 1, 2]))) == [0, 2, 3]


def test_yolo_decode_golden():
    g = np.load(os.path.join(GOLDEN, "yolo_layer.npz"))
    anchors = [g["anchors"][m].tolist() for m in g["mask"]]
    bb, p, c = orc.yolo_decode(torch.from_numpy(g["x"]), anchors)
    np.testing.assert_allclose(bb.numpy(), g["bbox"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(p.numpy(), g["prob"], rtol=1e-6, atol=1e-7)
    assert (c.numpy() == g["cls"]).all()
    bb, p, c = orc.yolo_decode(torch.from_numpy(g["probe"]), anchors)
    np.testing.assert_allclose(bb.numpy(), g["probe_bbox"], rtol=1e-6, atol=1e-6)
    row = 0 * 12 + 2 * 4 + 3
    assert int(c[0, row]) == 7 and c.dtype == torch.int64
    np.testing.assert_allclose(bb[0, row].numpy(), [(1 / (1 + np.exp(-1.0)) + 3) / 4, (0.5 + 2) / 3, 30.0, 61.0],
                               rtol=1e-6)


def test_mini_network_every_block():
    g = np.load(os.path.join(GOLDEN, "mini_blocks.npz"))
    cfg = os.path.join(GOLDEN, "cfg", "mini.cfg")
    net = orc.OracleDarknet(cfg)
    net.set_params(W.synth_params(net.blocks, net.net_info, seed=0, obj_bias=-5.0, calib=None))
    frames = synth_frames(7, 2, net.net_info["height"], net.net_info["width"], rects=12)
    assert (frames == g["frames"]).all()
    x = torch.from_numpy(orc.frames_to_input(list(frames)))
    np.testing.assert_array_equal(x.numpy(), g["input"])
    collect = {}
    out = net.forward(x, collect=collect)
    checked = 0
    for i, blk in enumerate(net.blocks):
        key = "block_%d" % i
        if key in g.files:
            np.testing.assert_allclose(collect[i].numpy(), g[key], rtol=2e-5, atol=2e-5, err_msg=key)
            checked += 1
    assert checked >= 22
    np.testing.assert_allclose(out["bbox_xywh"].numpy(), g["bbox_xywh"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["class_prob"].numpy(), g["class_prob"], rtol=1e-4, atol=1e-6)
    assert (out["class_idx"].numpy() == g["class_idx"]).mean() > 0.999


@pytest.mark.parametrize("model", ["yolov3-tiny", "yolov3", "yolov3-spp"])
def test_full_forward_golden(model, tmp_path):
    if model != "yolov3-tiny" and os.environ.get("Y3_FAST_CPU_TESTS"):
        pytest.skip("fast mode")
    g = np.load(os.path.join(GOLDEN, "forward_%s.npz" % model))
    dim = MODEL_DIMS[model]
    frames = [resize_bilinear_u8(load_jpeg_bgr("000000035279.jpg"), dim, dim), synth_frames(5, 1, dim, dim)[0]]
    assert [sha(f) for f in frames] == g["frames_sha"].tolist(), "input frames differ from the golden run"
    net = orc.OracleDarknet(MODELS[model]).load_weights(golden_weights_path(model, tmp_path))
    # one frame per model keeps the CPU suite short; frame 1 is the procedural scene
    idx = 1
    inp = orc.frames_to_input([frames[idx]])
    out = net.forward(torch.from_numpy(inp))
    assert out["bbox_xywh"].shape == (1, g["bbox_xywh"].shape[1], 4)
    np.testing.assert_allclose(out["bbox_xywh"].numpy()[0], g["bbox_xywh"][idx], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(out["class_prob"].numpy()[0], g["class_prob"][idx], atol=1e-4)
    same = out["class_idx"].numpy()[0] == g["class_idx"][idx]
    assert (same | (g["cls_margin"][idx] < 1e-4)).all()


def test_inference_golden_tiny(tmp_path):
    model = "yolov3-tiny"
    g = np.load(os.path.join(GOLDEN, "inference_%s.npz" % model))
    dim = MODEL_DIMS[model]
    frames = [load_jpeg_bgr("000000229358.jpg"), synth_frames(9, 1, dim, dim)[0], load_jpeg_bgr("000000393569.jpg")]
    assert [sha(f) for f in frames] == g["frames_sha"].tolist()
    net = orc.OracleDarknet(MODELS[model]).load_weights(golden_weights_path(model, tmp_path))
    resized = [resize_bilinear_u8(f, dim, dim) for f in frames]
    out = net.forward(torch.from_numpy(orc.frames_to_input(resized)))
    for tag in ("a", "b"):
        pth, ith = g[tag + "_thresholds"]
        res = orc.postprocess(out["bbox_xywh"].numpy(), out["class_prob"].numpy(), out["class_idx"].numpy(),
                              [f.shape for f in frames], float(pth), float(ith))
        for f in range(len(frames)):
            compare_detections(g, "%s_f%d_" % (tag, f), res[f], rows=None)
