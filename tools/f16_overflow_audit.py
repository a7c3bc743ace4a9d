#!/usr/bin/env python3
"""Overflow / underflow audit of the fp16 storage mode: the largest |value| every stored tensor takes
-- the normalised input, every conv weight tensor, every block output the fp16 plan keeps in HBM -- over the golden frames,
against the largest finite half (65 504), and how much of each tensor falls below the smallest NORMAL half (2^-14: stored as
subnormals, with fewer significand bits) or rounds to zero (below 2^-25).  CPU only: the fp16-emulating oracle
(oracle/darknet_oracle.py, ``emulate="f16"``), whose per-block outputs the HIP path matches to one fp16 ulp
(tests/test_gpu_bf16.py::test_fp16_every_block_teacher_forced).

    python tools/f16_overflow_audit.py > profiles/r05_f16_overflow_audit.txt
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (os.path.join(ROOT, "pytorch-yolov3_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

from oracle import darknet_oracle as orc  # noqa: E402
from golden_util import (BENCH_REGIME_OBJ_BIAS, MODELS, MODEL_DIMS, SAMPLE_IMAGES, golden_params, load_jpeg_bgr)  # noqa: E402
from yolov3.preprocess import resize_bilinear_u8  # noqa: E402
from yolov3.synthdata import synth_frames  # noqa: E402

F16_MAX, F16_MIN_NORMAL, F16_HALF_MIN_SUB = 65504.0, 2.0 ** -14, 2.0 ** -25


def audit(model, params, label):
    dim = MODEL_DIMS[model]
    net = orc.OracleDarknet(MODELS[model]).set_params(params)
    frames = [resize_bilinear_u8(load_jpeg_bgr(n), dim, dim) for n in SAMPLE_IMAGES] + list(synth_frames(9, 1, dim, dim)) + \
        list(synth_frames(1000, 2, dim, dim))
    rounds = net.bf16_rounding_points()
    n = len(net.blocks)
    mx = np.zeros(n)
    sub = np.zeros(n)
    zero = np.zeros(n)
    cnt = np.zeros(n)
    nonfinite = 0
    for f in frames:
        col = {}
        out = net.forward(torch.from_numpy(orc.frames_to_input([f])), emulate="f16", collect=col)
        nonfinite += int((~torch.isfinite(out["class_prob"])).sum()) + int((~torch.isfinite(out["bbox_xywh"])).sum())
        for i, t in col.items():
            if not torch.is_tensor(t) or net.blocks[i]["type"] == "yolo":
                continue
            a = t.abs()
            mx[i] = max(mx[i], float(a.max()))
            nz = a > 0
            sub[i] += float(((a < F16_MIN_NORMAL) & nz).sum())
            cnt[i] += a.numel()
            nonfinite += int((~torch.isfinite(t)).sum())
    wmax, wsub, wzero = 0.0, 0.0, 0.0
    wn = 0
    for p in params:
        w = np.abs(p["weight"].astype(np.float32))
        wmax = max(wmax, float(w.max()))
        wsub += float(((w < F16_MIN_NORMAL) & (w >= F16_HALF_MIN_SUB)).sum())
        wzero += float(((w < F16_HALF_MIN_SUB) & (w > 0)).sum())
        wn += w.size
    stored = [i for i in range(n) if net.blocks[i]["type"] in ("convolutional", "shortcut") and
              (net.blocks[i]["type"] == "shortcut" or rounds[i])]
    worst = max(stored, key=lambda i: mx[i])
    print("%-28s %2d frames  largest stored |activation| %8.2f at block %3d (%s)  = %.5f of 65504   non-finite values: %d" % (
        label, len(frames), mx[worst], worst, net.blocks[worst]["type"], mx[worst] / F16_MAX, nonfinite))
    print("%-28s weights: largest |w| %.3f; %.4f %% below 2^-14 (kept as subnormals), %.5f %% below 2^-25 (round to zero)" % (
        "", wmax, 100 * wsub / wn, 100 * wzero / wn))
    top = sorted(stored, key=lambda i: -mx[i])[:6]
    print("%-28s six largest blocks: %s" % ("", ", ".join("%d: %.1f" % (i, mx[i]) for i in top)))
    frac_sub = sum(sub[i] for i in stored) / max(1.0, sum(cnt[i] for i in stored))
    print("%-28s stored activations below 2^-14 in magnitude (non-zero): %.4f %%" % ("", 100 * frac_sub))
    return mx[worst], nonfinite


def main():
    from yolov3 import weights as W
    from yolov3.cfgparse import parse_config
    print("fp16 storage mode: range audit on the golden frames (nine sample images + three procedural frames per model),")
    print("oracle emulate=\"f16\" (CPU); largest finite half = 65504, smallest normal half = 6.1e-5\n")
    worst, bad = 0.0, 0
    for model in ("yolov3-tiny", "yolov3", "yolov3-spp"):
        for label, params in (("%s golden (obj bias -5)" % model, golden_params(model)),
                              ("%s bench regime (%.1f)" % (model, BENCH_REGIME_OBJ_BIAS[model]), golden_params(model, BENCH_REGIME_OBJ_BIAS[model]))):
            m, nf = audit(model, params, label)
            worst, bad = max(worst, m), bad + nf
    blocks, net_info = parse_config(MODELS["yolov3"])
    m, nf = audit("yolov3", W.planted_params(blocks, net_info), "yolov3 planted head")
    worst, bad = max(worst, m), bad + nf
    print("\nlargest stored value anywhere: %.2f = %.5f of the half range (headroom %.0fx); non-finite values: %d -> %s" % (
        worst, worst / F16_MAX, F16_MAX / worst, bad, "NO overflow" if bad == 0 and worst < F16_MAX / 4 else "CHECK"))


if __name__ == "__main__":
    main()
