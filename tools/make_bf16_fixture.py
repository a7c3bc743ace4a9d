#!/usr/bin/env python3
"""tests/golden/bf16_agreement.json: what an IDEAL bf16 implementation reaches against the reference's float32
``inference()`` lists (G7, tests/golden/inference_*.npz, produced by the reference itself).

The reference has no bf16 mode; the bf16-emulating oracle (oracle/darknet_oracle.py, ``emulate_bf16=True``) stands for
"a correct bf16 implementation".  For every G7 frame and threshold pair this stores the keep-set Jaccard (by prediction
row) and the score differences on the common rows, for the oracle accumulating in float32 and in float64 (the spread
between the two is the summation-order noise).  tests/test_gpu_bf16.py and bench.py (``bf16_agreement``) compare the HIP
bf16 path with these floors.  Runs on CPU, needs nothing outside the repository:  python tools/make_bf16_fixture.py

``--emulate f16`` (round 5) writes tests/golden/f16_agreement.json instead: the same table for the fp16 storage mode
(``OracleDarknet.forward(emulate="f16")``), the floors of the ``dtype="fp16"`` HIP path.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (os.path.join(ROOT, "pytorch-yolov3_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

from oracle import darknet_oracle as orc  # noqa: E402
from golden_util import (BENCH_REGIME_OBJ_BIAS, GOLDEN, MODELS, MODEL_DIMS, bench_regime_frame, golden_params,  # noqa: E402
                         load_jpeg_bgr)
from yolov3.preprocess import resize_bilinear_u8  # noqa: E402
from yolov3.synthdata import synth_frames  # noqa: E402


def agreement(det, g, prefix):
    rows = set(int(r) for r in det[3])
    want = set(g[prefix + "rows"].tolist())
    gp = dict(zip(g[prefix + "rows"].tolist(), g[prefix + "prob"].tolist()))
    mine = {int(r): k for k, r in enumerate(det[3])}
    common = sorted(rows & want)
    dp = np.array([abs(float(det[1][mine[r]]) - gp[r]) for r in common]) if common else np.zeros(1)
    return dict(jaccard=round(len(rows & want) / len(rows | want), 4) if rows | want else 1.0, kept=len(rows), ref_kept=len(want),
                score_med=float(np.median(dp)), score_p99=float(np.percentile(dp, 99)), score_max=float(dp.max()))


def bench_regime():
    """The same floors at the BENCHMARKED regime (objectness bias -8.5 / -5.0: tens of kept boxes per frame, like trained
    weights) on every frame of tests/golden/inference_bench_regime_<model>.npz: the nine sample images and the procedural
    frames, one image per call.  ``all`` pools the frames: |intersection| / |union| of kept rows summed over frames."""
    out = {}
    for model in ("yolov3-tiny", "yolov3", "yolov3-spp"):
        g = np.load(os.path.join(GOLDEN, "inference_bench_regime_%s.npz" % model))
        dim = MODEL_DIMS[model]
        net = orc.OracleDarknet(MODELS[model]).set_params(golden_params(model, BENCH_REGIME_OBJ_BIAS[model]))
        entry = {}
        pool = {t: [0, 0] for t in ("a", "b")}
        for name in (str(n) for n in g["names"]):
            frame = bench_regime_frame(name, dim)
            x = torch.from_numpy(orc.frames_to_input([resize_bilinear_u8(frame, dim, dim)]))
            o = net.forward(x, emulate=EMULATE, accumulate="f32")
            for tag in ("a", "b"):
                pth, ith = g[tag + "_thresholds"]
                det = orc.postprocess(o["bbox_xywh"].numpy(), o["class_prob"].numpy(), o["class_idx"].numpy(), [frame.shape],
                                      float(pth), float(ith), audit=True)[0]
                key = "%s_%s" % (name, tag)
                entry[key] = agreement(det, g, key + "_")
                rows, want = set(int(r) for r in det[3]), set(g[key + "_rows"].tolist())
                pool[tag][0] += len(rows & want)
                pool[tag][1] += len(rows | want)
                print(model, key, entry[key])
        for tag in ("a", "b"):
            entry["all_" + tag] = dict(jaccard=round(pool[tag][0] / max(pool[tag][1], 1), 4), common=pool[tag][0], union=pool[tag][1])
            print(model, "all", tag, entry["all_" + tag])
        out[model] = entry
    return out


def planted():
    """Floors on the PLANTED parameter set (tools/make_planted.py: a fitted 19 x 19 head, a handful of confident detections per
    sample image with margins): keep-set Jaccard of the bf16-emulating oracle against the reference's float32 lists, pooled
    over the nine frames, and whether the classes agree on the common rows."""
    from yolov3 import weights as W
    from yolov3.cfgparse import parse_config
    g = np.load(os.path.join(GOLDEN, "inference_planted_yolov3.npz"))
    blocks, net_info = parse_config(MODELS["yolov3"])
    net = orc.OracleDarknet(MODELS["yolov3"]).set_params(W.planted_params(blocks, net_info))
    entry = {}
    pool = {t: [0, 0, 0, 0] for t in ("a", "b")}
    for name in (str(n) for n in g["names"]):
        frame = resize_bilinear_u8(load_jpeg_bgr("000000%s.jpg" % name), 608, 608)
        o = net.forward(torch.from_numpy(orc.frames_to_input([frame])), emulate=EMULATE, accumulate="f32")
        for tag in ("a", "b"):
            pth, ith = g[tag + "_thresholds"]
            det = orc.postprocess(o["bbox_xywh"].numpy(), o["class_prob"].numpy(), o["class_idx"].numpy(), [frame.shape],
                                  float(pth), float(ith), audit=True)[0]
            key = "%s_%s" % (name, tag)
            entry[key] = agreement(det, g, key + "_")
            rows, want = [int(r) for r in det[3]], g[key + "_rows"].tolist()
            gcls = dict(zip(want, g[key + "_cls"].tolist()))
            same_cls = sum(1 for r, c in zip(rows, det[2]) if r in gcls and gcls[r] == int(c))
            pool[tag][0] += len(set(rows) & set(want))
            pool[tag][1] += len(set(rows) | set(want))
            pool[tag][2] += same_cls
            pool[tag][3] = max(pool[tag][3], len(want))
            print("planted", key, entry[key])
    for tag in ("a", "b"):
        entry["all_" + tag] = dict(jaccard=round(pool[tag][0] / max(pool[tag][1], 1), 4), common=pool[tag][0], union=pool[tag][1],
                                   same_class_on_common=pool[tag][2], most_kept_per_frame=pool[tag][3])
        print("planted all", tag, entry["all_" + tag])
    return entry


EMULATE = "bf16"


def main():
    global EMULATE
    argv = sys.argv[1:]
    if argv[:1] == ["--emulate"]:
        EMULATE = argv[1]
        argv = argv[2:]
    assert EMULATE in ("bf16", "f16")
    path = os.path.join(GOLDEN, "%s_agreement.json" % EMULATE)
    if argv == ["planted"]:                    # only that section (the rest takes minutes)
        with open(path) as fh:
            table = json.load(fh)
        table["planted"] = {"yolov3": planted()}
        with open(path, "w") as fh:
            json.dump(table, fh, indent=1, sort_keys=True)
        return
    table = {}
    for model in ("yolov3-tiny", "yolov3", "yolov3-spp"):
        g = np.load(os.path.join(GOLDEN, "inference_%s.npz" % model))
        dim = MODEL_DIMS[model]
        frames = [load_jpeg_bgr("000000229358.jpg"), synth_frames(9, 1, dim, dim)[0], load_jpeg_bgr("000000393569.jpg")]
        x = torch.from_numpy(orc.frames_to_input([resize_bilinear_u8(f, dim, dim) for f in frames]))
        net = orc.OracleDarknet(MODELS[model]).set_params(golden_params(model))
        entry = {}
        for acc in ("f32", "f64"):
            o = net.forward(x, emulate=EMULATE, accumulate=acc)
            for tag in ("a", "b"):
                pth, ith = g[tag + "_thresholds"]
                dets = orc.postprocess(o["bbox_xywh"].numpy(), o["class_prob"].numpy(), o["class_idx"].numpy(),
                                       [f.shape for f in frames], float(pth), float(ith), audit=True)
                for f in range(len(frames)):
                    a = agreement(dets[f], g, "%s_f%d_" % (tag, f))
                    key = "%s_f%d" % (tag, f)
                    if acc == "f32":
                        entry[key] = a
                    else:
                        entry[key]["jaccard_f64acc"] = a["jaccard"]
                    print(model, acc, key, a)
        table[model] = entry
    table["bench_regime"] = bench_regime()
    table["planted"] = {"yolov3": planted()}
    with open(path, "w") as fh:
        json.dump(table, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
