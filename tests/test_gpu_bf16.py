"""bf16 / fp16 parity gates (-m gpu): the 16-bit storage modes against the oracle emulating that storage type.

Every gate below exists for both modes (``MODES``): bf16 is the benchmarked dtype (BASELINE.json configs[2]); fp16 is the
same kernels instantiated on ``v_mfma_f32_16x16x32_f16`` (round 5: same rate and bytes, 11 significand bits instead of 8).
The text speaks of bf16; for fp16 read "IEEE half" and an ulp of 2^-10 relative instead of 2^-7.

Three levels, from tight to loose:

1. ``test_bf16_every_block_teacher_forced``: every block of a network, each fed with the PRODUCT's own input
   tensor for that block, against the oracle's bf16 emulation of that one block (``oracle/darknet_oracle.py``:
   bf16 weights and inputs, float32 accumulate / BatchNorm / LeakyReLU / shortcut add, one rounding where the
   tensor is stored).  With identical inputs the only legitimate difference is float32 summation order, i.e. at
   most ONE bf16 ulp on the few values that sit on a rounding boundary -- a wrong border column, tap, K tile or
   channel slice is O(1).  Covers every conv kernel family the plans select (MFMA stem, fused first two convs,
   fused residual block, halo / patch / implicit-GEMM kernels, fused head conv + decode), pools, upsample, routes.
2. ``test_bf16_whole_net_vs_bf16_oracle``: end-to-end outputs against the emulating oracle, with tolerances
   derived in the test from the summation-order noise floor (oracle accumulating in float32 vs float64: through
   75 layers of procedural weights one-ulp flips amplify, so two CORRECT bf16 implementations differ by this much).
3. ``test_bf16_post_nms_agreement_vs_reference``: detections of the bf16 path against the reference's own
   ``inference()`` lists (G7): keep-set Jaccard and score differences, asserted against the floor an ideal bf16
   implementation reaches (tests/golden/bf16_agreement.json, tools/make_bf16_fixture.py).
"""
import json
import os

import numpy as np
import pytest
import torch

import yolov3
from oracle import darknet_oracle as orc
from yolov3.preprocess import resize_bilinear_u8
from yolov3.synthdata import synth_frames

from golden_util import (GOLDEN, MODELS, MODEL_DIMS, bf16_agreement, golden_params, golden_weights_path, load_jpeg_bgr)

pytestmark = pytest.mark.gpu

BF16_ULP_REL = 2.0 ** -7        # spacing of bf16 values relative to their magnitude is in (2^-8, 2^-7]
MISMATCH_MAX = 0.02             # share of values allowed to land on the other side of a rounding boundary
# per storage mode: product dtype, oracle emulation, rounding function, ulp relative to the magnitude, significand bits,
# smallest spacing (subnormals), the tag kernel names carry
MODES = {
    "bf16": dict(dtype="bf16", emulate="bf16", rnd=orc.bf16_round, ulp_rel=2.0 ** -7, bits=8, min_ulp=0.0, tag="bf16",
                 # post-NMS score slack: added to 2 x the ideal median / p99; max of a handful of scores; bench-regime median
                 # and max caps; planted-set score and box (px) caps
                 med_add=1e-3, p99_add=5e-3, few_max=3e-2, regime_med=1e-2, regime_max=6e-2, planted_score=0.1, planted_box=6),
    "fp16": dict(dtype="fp16", emulate="f16", rnd=orc.f16_round, ulp_rel=2.0 ** -10, bits=11, min_ulp=2.0 ** -24, tag="f16",
                 med_add=2e-4, p99_add=1e-3, few_max=5e-3, regime_med=1.5e-3, regime_max=1e-2, planted_score=0.02, planted_box=2),
}


def _net(model, mode="bf16", **kw):
    net = yolov3.Darknet(MODELS[model], device="cuda", dtype=MODES[mode]["dtype"], **kw)
    if model == "mini":
        net.set_params(golden_params("mini"))
    else:
        net.load_weights(golden_weights_path(model))
    return net.eval()


def _bf16_ulp(r, mode="bf16"):
    """Spacing of the storage type's values at the (storage-valued) tensor r: 2^(exponent - 7) for bf16, 2^(exponent - 10)
    for fp16 (not below its subnormal spacing 2^-24)."""
    _, e = torch.frexp(r)                      # r = m * 2^e, m in [0.5, 1)
    ulp = torch.where(r == 0, torch.zeros_like(r), torch.ldexp(torch.ones_like(r), e - MODES[mode]["bits"]))
    return torch.clamp(ulp, min=MODES[mode]["min_ulp"]) if MODES[mode]["min_ulp"] else ulp


def _flip_slack(mid, w2, alpha2, stride, pad, mode="bf16"):
    """A fused conv pair keeps its intermediate tensor on chip, rounded to bf16: where the oracle's unrounded value
    ``mid`` sits on a bf16 rounding boundary (within float32 summation noise), the kernel may legitimately hold the
    neighbouring bf16 value.  Returns, per OUTPUT element of the second conv, how far such flips in its receptive
    field can move it: sum over taps of |w2| * |BN scale| * ulp(mid) over the boundary candidates (LeakyReLU's
    slope is <= 1).  Zero for outputs that see no candidate."""
    r = MODES[mode]["rnd"](mid)
    ulp = _bf16_ulp(r, mode)
    rms = float(mid.pow(2).mean().sqrt())
    dist = 0.5 * ulp - (mid - r).abs()         # distance of mid to the nearer rounding boundary
    cand = (dist <= 1e-5 * (mid.abs() + rms)).float()
    moved = torch.nn.functional.conv2d(cand * 2.0 * ulp, w2.abs(), None, stride=stride, padding=pad)
    return moved * alpha2.abs().reshape(1, -1, 1, 1)


def _close_bf16(got, want, what, slack=None, mode="bf16"):
    """``got``/``want``: float32 tensors holding bf16 values.  One bf16 ulp + a floor for values near zero
    (+ ``slack``, see ``_flip_slack``)."""
    got, want = got.float(), want.float()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    rms = float(want.pow(2).mean().sqrt())
    d = (got - want).abs()
    tol = MODES[mode]["ulp_rel"] * want.abs() + 1e-4 * rms + 1e-30
    if slack is not None:
        tol = tol + slack
        d = torch.where(slack > 0, torch.minimum(d, tol), d)      # counted as explained below, still bounded above
    worst = float((d / tol).max())
    frac = float(((d > 0) & ((slack == 0) if slack is not None else True)).float().mean())
    assert worst <= 1.0, "%s: |d| up to %.2f x (one %s ulp), max |d| %.3g, rms %.3g" % (what, worst, mode, float(d.max()), rms)
    assert frac <= MISMATCH_MAX, "%s: %.2f %% of the values differ (summation order explains < %.0f %%)" % (
        what, 100 * frac, 100 * MISMATCH_MAX)
    return frac


def _teacher_forced(model, frames, expect_kernels=(), options=None, mode="bf16", frames_checked=None):
    """``frames_checked``: the frames of the batch whose blocks are compared with the oracle (the product runs the WHOLE batch, so the plan
    and every kernel's grid are the batch's; the CPU oracle, 95 % of this test's time at batch 16, emulates only these).  None = all."""
    rnd, emulate = MODES[mode]["rnd"], MODES[mode]["emulate"]
    expect_kernels = tuple(k.replace("bf16", MODES[mode]["tag"]) for k in expect_kernels)
    net = _net(model, mode, keep_all=True, fuse=True, options=options)
    out = net.forward_frames(frames)
    torch.cuda.synchronize()
    report = net.plan_report()
    kernel_of = {}
    for r in report:
        kernel_of.setdefault(r["block"], []).append(r["kernel"])
    names = [r["kernel"] for r in report]
    for frag in expect_kernels:
        assert any(frag in k for k in names), "no %s kernel in the plan: %s" % (frag, sorted(set(names)))

    ref = orc.OracleDarknet(MODELS[model]).set_params(net._params)
    blocks = ref.blocks
    rounds = ref.bf16_rounding_points()
    sel = list(range(len(frames))) if frames_checked is None else sorted(set(frames_checked))
    x_net = rnd(torch.from_numpy(orc.frames_to_input([frames[j] for j in sel])))

    def hip(i):
        return x_net if i < 0 else net.block_output(i)[sel].cpu()

    def conv(i, x):
        blk = blocks[i]
        k = blk["size"]
        pad = (k - 1) // 2 if "pad" in blk else 0
        return orc.conv_block(x, ref.params[ref._conv_slot[i]], blk["stride"], pad, blk["activation"] == "leaky",
                              round_weights=emulate)

    def fused_away(i):
        return kernel_of.get(i, [""])[0].startswith("(fused")

    checked, worst_frac = 0, 0.0
    heads = []
    for i, blk in enumerate(blocks):
        kind = blk["type"]
        what = "%s block %d (%s, %s)" % (model, i, kind, ",".join(kernel_of.get(i, ["-"])))
        if kind == "convolutional":
            if i + 1 < len(blocks) and fused_away(i + 1) and blocks[i + 1]["type"] == "convolutional":
                continue                                   # first half of a fused pair: checked with its second half
            slack = None
            if fused_away(i) and blocks[i - 1]["type"] == "convolutional":
                mid = conv(i - 1, hip(i - 2))              # the pair's intermediate tensor lives in LDS as bf16
                x = rnd(mid)
                p2 = ref.params[ref._conv_slot[i]]
                alpha2 = torch.from_numpy(p2["bn_gamma"] / np.sqrt(p2["bn_var"] + orc.BN_EPS))
                k = blk["size"]
                slack = _flip_slack(mid, rnd(torch.from_numpy(p2["weight"])), alpha2, blk["stride"],
                                    (k - 1) // 2 if "pad" in blk else 0, mode)
            else:
                x = hip(i - 1)
            y = conv(i, x)
            nxt = blocks[i + 1]["type"] if i + 1 < len(blocks) else None
            if nxt == "yolo":
                heads.append((i + 1, y))                   # float32 logits: checked through the decode below
                continue
            if not rounds[i]:                              # conv + shortcut in one epilogue, one rounding of the sum
                sc = i + 1
                y = y + hip(sc + blocks[sc]["from"])
                worst_frac = max(worst_frac, _close_bf16(hip(sc), rnd(y), what + " + shortcut", slack, mode))
            else:
                worst_frac = max(worst_frac, _close_bf16(hip(i), rnd(y), what, slack, mode))
            checked += 1
        elif kind == "shortcut":
            if not rounds[i - 1]:
                continue                                   # checked with its conv
            want = rnd(hip(i - 1) + hip(i + blk["from"]))
            _close_bf16(hip(i), want, what, None, mode)
            checked += 1
        elif kind == "maxpool":
            assert torch.equal(hip(i), orc.maxpool(hip(i - 1), blk["size"], blk["stride"])), what
            checked += 1
        elif kind == "upsample":
            assert torch.equal(hip(i), orc.upsample(hip(i - 1), blk["stride"])), what
            checked += 1
        elif kind == "route":
            assert torch.equal(hip(i), torch.cat([hip(j) for j in blk["layers"]], dim=1)), what
            checked += 1
    # detection heads: decode of the oracle's float32 logits (from the product's head-conv input) against the
    # rows this head wrote into the final outputs
    bb = out["bbox_xywh"][sel].cpu()
    pr = out["class_prob"][sel].cpu()
    ci = out["class_idx"][sel].cpu()
    row = 0
    for yi, logits in heads:
        blk = blocks[yi]
        mask = blk["mask"] if isinstance(blk["mask"], list) else [blk["mask"]]
        box, prob, idx = orc.yolo_decode(logits, [blk["anchors"][m] for m in mask])
        box[:, :, 2] /= ref.net_info["width"]
        box[:, :, 3] /= ref.net_info["height"]
        n = prob.shape[1]
        what = "%s head at block %d (%s)" % (model, yi, ",".join(kernel_of.get(yi - 1, ["-"])))
        torch.testing.assert_close(bb[:, row:row + n], box, rtol=2e-4, atol=2e-5, msg=lambda m: what + " boxes: " + m)
        torch.testing.assert_close(pr[:, row:row + n], prob, rtol=5e-4, atol=2e-5, msg=lambda m: what + " scores: " + m)
        top2 = torch.softmax(logits.reshape(logits.shape[0], len(mask), -1, logits.shape[2], logits.shape[3])[:, :, 5:], dim=2)
        top2 = torch.topk(top2, 2, dim=2).values
        margin = (top2[:, :, 0] - top2[:, :, 1]).reshape(logits.shape[0], -1)
        flips = (ci[:, row:row + n] != idx) & (margin > 1e-3)
        assert int(flips.sum()) == 0, what + ": arg-max flips on clear margins"
        row += n
        checked += 1
    assert row == pr.shape[1]
    return checked, worst_frac, names


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_bf16_every_block_teacher_forced_mini(mode):
    g = np.load(os.path.join(GOLDEN, "mini_blocks.npz"))
    checked, frac, names = _teacher_forced("mini", g["frames"], mode=mode)
    print("mini: %d blocks checked, worst mismatch share %.4f; kernels %s" % (checked, frac, sorted(set(names))))
    assert checked >= 20


from yolov3 import _hip as _H
HALO = {"auto_mask": _H.AM_DEFAULT | _H.AM_NO_SMALL_GRID}     # round-1 selection: the halo kernel whatever the grid size
NO_SMALL_DW = {"auto_mask": _H.AM_DEFAULT & ~_H.AM_SMALL_DW}  # round-5 selection for small grids: the wave-specialised implicit GEMM
WRES_ALWAYS = {"auto_mask": (_H.AM_DEFAULT & ~_H.AM_SMALL_DW) | _H.AM_WRES_ALWAYS}


@pytest.mark.parametrize("model,h,w,batch,options,kernels", [
    ("yolov3-tiny", 416, 416, 2, None, ("conv_stem_mfma", "conv_igemm", "head_decode", "maxpool")),
    # default selection at these small batches (round 6): the small-grid direct-weights kernel, 1x1 and 3x3
    ("yolov3", 608, 608, 1, None, ("conv_stem_s2_fused", "conv_resblock_fused", "conv_patch", "conv_igemm2", "conv_dw48_k3", "conv_dw48_k1",
                                   "head_decode")),
    ("yolov3", 352, 480, 2, None, ("conv_stem_s2_fused", "conv_resblock_fused", "conv_dw48_k3", "conv_dw48_k1", "head_decode")),
    ("yolov3-spp", 416, 416, 2, None, ("conv_dw48_k3", "conv_dw48_k1", "maxpool_spp", "head_decode")),
    # ... and without it (round 5's selection): the 3x3 layers with few tiles on the wave-specialised implicit GEMM
    ("yolov3", 608, 608, 1, NO_SMALL_DW, ("conv_stem_s2_fused", "conv_resblock_fused", "conv_patch", "conv_igemm2", "conv_igemm3",
                                          "head_decode")),
    ("yolov3", 352, 480, 2, NO_SMALL_DW, ("conv_stem_s2_fused", "conv_resblock_fused", "conv_igemm3", "head_decode")),
    ("yolov3-spp", 416, 416, 2, NO_SMALL_DW, ("conv_igemm3", "maxpool_spp", "head_decode")),
    # the halo kernel on every layer it fits (what the batch-16 benchmark runs), 192-pixel tiles at these sizes
    ("yolov3", 608, 608, 1, HALO, ("conv_stem_s2_fused", "conv_resblock_fused", "conv_halo_ws", "conv_patch", "conv_igemm2",
                                   "conv_igemm3", "head_decode")),
    ("yolov3", 352, 480, 2, HALO, ("conv_stem_s2_fused", "conv_resblock_fused", "conv_halo_ws", "head_decode")),
    ("yolov3", 320, 320, 3, HALO, ("conv_halo_ws", "head_decode")),
    ("yolov3-spp", 608, 608, 1, HALO, ("conv_halo_ws", "maxpool_spp", "head_decode")),
    ("yolov3-spp", 416, 416, 2, HALO, ("conv_halo_ws", "maxpool_spp", "head_decode")),
    # batch 8: 76^2 and 38^2 stay on the halo kernel by themselves, 19^2 goes to the small-grid kernel (round 5: the implicit GEMM)
    ("yolov3", 608, 608, 8, None, ("conv_halo_ws", "conv_dw48_k3", "conv1x1_dw", "head_decode")),
    ("yolov3", 608, 608, 8, NO_SMALL_DW, ("conv_halo_ws", "conv_igemm3", "head_decode")),
    # the weights-resident persistent 1x1 kernel on every layer it supports (Y3_AM_WRES_ALWAYS; by itself it starts at 512 tiles)
    ("yolov3", 608, 608, 2, WRES_ALWAYS, ("conv1x1_wres_bf16_128x128", "conv1x1_wres_bf16_128x64")),
    ("yolov3", 352, 480, 3, WRES_ALWAYS, ("conv1x1_wres_bf16_128x128", "conv1x1_wres_bf16_128x64")),
    ("yolov3-spp", 416, 416, 1, WRES_ALWAYS, ("conv1x1_wres",)),
    # THE BENCHMARKED CONFIGURATION, block by block: 16 frames of 608 x 608 with bench.py's plan options (256-pixel halo
    # tiles in several rounds of workgroups, the direct-weights strip and 1x1 kernels, the weights-resident 1x1 kernel by itself, the
    # stride-2 layer on igemm3)
    ("yolov3", 608, 608, 16, {"auto_mask": _H.AM_DEFAULT | _H.AM_HALO_TILE256}, ("conv_halo_ws_bf16_256x128", "conv_halo_dw_bf16_192x256",
                                                        "conv1x1_dw_bf16_96x256", "conv1x1_dw_bf16_48x256", "conv1x1_wres_bf16_128x128",
                                                        "conv1x1_wres_bf16_128x64", "conv_igemm3", "conv_patch", "head_decode")),
])
def test_bf16_every_block_teacher_forced(model, h, w, batch, options, kernels):
    _every_block(model, h, w, batch, options, kernels, "bf16")


def _every_block(model, h, w, batch, options, kernels, mode):
    frames = synth_frames(1000 + h + w + batch, batch, h, w)
    if (h, w) == (608, 608):
        frames[0] = resize_bilinear_u8(load_jpeg_bgr("000000035279.jpg"), h, w)
    # batches of 8 / 16: the first, a middle and the last frame against the oracle (tiles of the raster strips straddle frames; that every
    # frame of a full-size batch equals the same frame alone is test_frames_are_independent_at_full_size's job)
    some = (0, batch // 2, batch - 1) if batch >= 8 else None
    checked, frac, names = _teacher_forced(model, frames, kernels, options=options, mode=mode, frames_checked=some)
    print("%s %s %dx%d b%d %s: %d blocks checked, worst mismatch share %.4f" % (mode, model, h, w, batch, options, checked, frac))
    assert checked >= (20 if model == "yolov3-tiny" else 75)      # 107 blocks, 23 convs checked with their shortcut, 3 yolo


# fp16 (round 5): the same per-block gate at one fp16 ulp on every kernel family -- the MFMA stem, both fused early kernels,
# strip kernel with 192- and 256-pixel tiles, patch kernel, the three implicit GEMMs, the weights-resident 1x1, fused head
# conv + decode, the SPP pyramid -- and on THE BENCHMARKED PLAN (16 frames of 608 x 608, bench.py's options)
@pytest.mark.parametrize("model,h,w,batch,options,kernels", [
    ("yolov3-tiny", 416, 416, 2, None, ("conv_stem_mfma", "conv_igemm", "head_decode", "maxpool")),
    ("yolov3", 608, 608, 1, None, ("conv_stem_s2_fused", "conv_resblock_fused", "conv_patch", "conv_igemm2", "conv_dw48_k3", "conv_dw48_k1",
                                   "head_decode")),
    ("yolov3", 608, 608, 1, NO_SMALL_DW, ("conv_stem_s2_fused", "conv_resblock_fused", "conv_patch", "conv_igemm2", "conv_igemm3",
                                          "head_decode")),
    ("yolov3", 352, 480, 2, HALO, ("conv_stem_s2_fused", "conv_resblock_fused", "conv_halo_ws", "head_decode")),
    ("yolov3-spp", 608, 608, 1, HALO, ("conv_halo_ws", "maxpool_spp", "head_decode")),
    ("yolov3", 608, 608, 2, WRES_ALWAYS, ("conv1x1_wres_bf16_128x128", "conv1x1_wres_bf16_128x64")),
    ("yolov3", 608, 608, 16, {"auto_mask": _H.AM_DEFAULT | _H.AM_HALO_TILE256}, ("conv_halo_ws_bf16_256x128", "conv_halo_dw_bf16_192x256",
                                                        "conv1x1_dw_bf16_96x256", "conv1x1_dw_bf16_48x256", "conv1x1_wres_bf16_128x128",
                                                        "conv1x1_wres_bf16_128x64", "conv_igemm3", "conv_patch", "head_decode")),
])
def test_fp16_every_block_teacher_forced(model, h, w, batch, options, kernels):
    _every_block(model, h, w, batch, options, kernels, "fp16")


@pytest.mark.parametrize("model,h,w,batch", [("yolov3", 608, 608, 1), ("yolov3", 320, 416, 2)])
def test_bf16_every_block_teacher_forced_256_pixel_halo_tiles(model, h, w, batch):
    """The halo kernel picks 192- or 256-pixel tiles per layer from the tile count (192 at these small batches); the same
    per-block gate with 256-pixel tiles forced (Y3_AM_HALO_TILE256), the choice the batch-16 benchmark makes at 76^2."""
    frames = synth_frames(3000 + h + w + batch, batch, h, w)
    checked, frac, names = _teacher_forced(model, frames, ("conv_halo_ws_bf16_256x128",), options={"auto_mask": _H.AM_DEFAULT | _H.AM_HALO_TILE256 | _H.AM_NO_SMALL_GRID})
    assert not any("192x128" in k for k in names)
    print("%s %dx%d b%d: %d blocks checked, worst mismatch share %.4f" % (model, h, w, batch, checked, frac))


def _stats(a, b):
    d = np.abs(a["class_prob"] - b["class_prob"])
    rel = np.abs(a["bbox_xywh"] - b["bbox_xywh"]) / (np.abs(b["bbox_xywh"]) + 1e-6)
    return dict(score_med=float(np.median(d)), score_p99=float(np.percentile(d, 99)), score_max=float(d.max()),
                box_rel_p99=float(np.percentile(rel, 99)), argmax_disagree=float((a["class_idx"] != b["class_idx"]).mean()))


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("model", ["yolov3-tiny", "yolov3", "yolov3-spp"])
def test_bf16_whole_net_vs_bf16_oracle(model, mode):
    dim = MODEL_DIMS[model]
    frames = np.stack([resize_bilinear_u8(load_jpeg_bgr("000000035279.jpg"), dim, dim), synth_frames(5, 1, dim, dim)[0]])
    out = _net(model, mode).forward_frames(frames)
    got = {k: v.cpu().numpy() for k, v in out.items()}
    ref = orc.OracleDarknet(MODELS[model]).set_params(golden_params(model))
    x = torch.from_numpy(orc.frames_to_input(list(frames)))
    o32 = {k: v.numpy() for k, v in ref.forward(x, emulate=MODES[mode]["emulate"]).items()}
    o64 = {k: v.numpy() for k, v in ref.forward(x, emulate=MODES[mode]["emulate"], accumulate="f64").items()}
    floor = _stats(o32, o64)          # what summation order alone does to a correct bf16 implementation
    ours = _stats(got, o32)
    print(model, mode, "noise floor", floor)
    print(model, mode, "HIP vs oracle", ours)
    assert np.isfinite(got["bbox_xywh"]).all() and np.isfinite(got["class_prob"]).all()
    for key in ("score_med", "score_p99", "score_max", "box_rel_p99", "argmax_disagree"):
        assert ours[key] <= 3.0 * floor[key] + 1e-6, "%s: %s %.3g vs noise floor %.3g" % (model, key, ours[key], floor[key])


def _keep_agreement(det, g, prefix):
    rows = set(int(r) for r in det[3])
    want = set(g[prefix + "rows"].tolist())
    gp = dict(zip(g[prefix + "rows"].tolist(), g[prefix + "prob"].tolist()))
    gc = dict(zip(g[prefix + "rows"].tolist(), g[prefix + "cls"].tolist()))
    mine = {int(r): k for k, r in enumerate(det[3])}
    common = sorted(rows & want)
    dp = np.array([abs(float(det[1][mine[r]]) - gp[r]) for r in common]) if common else np.zeros(1)
    cls_same = float(np.mean([int(det[2][mine[r]]) == gc[r] for r in common])) if common else 1.0
    return (len(rows & want) / len(rows | want) if rows | want else 1.0), dp, cls_same, len(rows ^ want)


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("model", ["yolov3-tiny", "yolov3", "yolov3-spp"])
def test_bf16_post_nms_agreement_vs_reference(model, mode):
    """bf16 detections against the reference's float32 ``inference()`` lists (G7).  With procedural weights and
    thousands of overlapping near-threshold boxes per frame the keep set is sensitive to bf16 rounding as such:
    the floor is what the emulating oracle (an ideal bf16 implementation) reaches on the same frames."""
    g = np.load(os.path.join(GOLDEN, "inference_%s.npz" % model))
    M = MODES[mode]
    fixture = bf16_agreement(M["emulate"])[model]
    dim = MODEL_DIMS[model]
    frames = [load_jpeg_bgr("000000229358.jpg"), synth_frames(9, 1, dim, dim)[0], load_jpeg_bgr("000000393569.jpg")]
    net = _net(model, mode)
    for tag in ("a", "b"):
        pth, ith = g[tag + "_thresholds"]
        res = yolov3.inference(net, frames, device="cuda", prob_thresh=float(pth), nms_iou_thresh=float(ith),
                               return_rows=True)
        for f in range(len(frames)):
            jac, dp, cls_same, nxor = _keep_agreement(res[f], g, "%s_f%d_" % (tag, f))
            fl = fixture["%s_f%d" % (tag, f)]
            print("%s %s %s frame %d: keep-set Jaccard %.3f (ideal: %.3f), score |d| median %.1e p99 %.1e (ideal %.1e / %.1e), "
                  "class agreement on common rows %.4f" % (mode, model, tag, f, jac, fl["jaccard"], np.median(dp),
                                                          np.percentile(dp, 99), fl["score_med"], fl["score_p99"], cls_same))
            if fl["ref_kept"] >= 100:
                assert jac >= fl["jaccard"] - 0.08
            else:           # a handful of detections: count rows instead of a ratio
                ideal_xor = round((1.0 - fl["jaccard"]) * max(fl["kept"], fl["ref_kept"]) * 2)
                assert nxor <= ideal_xor + 3
            if len(dp) >= 20:
                assert np.median(dp) <= 2.0 * fl["score_med"] + M["med_add"] and np.percentile(dp, 99) <= 2.0 * fl["score_p99"] + M["p99_add"]
            else:           # a handful of scores: no percentiles, the bf16 score error as such (p99 ~ 2e-2 on these networks)
                assert dp.max() <= M["few_max"]
            assert cls_same >= 0.99


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("model", ["yolov3-tiny", "yolov3", "yolov3-spp"])
def test_bf16_post_nms_agreement_at_the_bench_regime(model, mode):
    """The same comparison at the objectness bias the throughput is measured at (tens of kept boxes per frame), on the nine
    sample images and the procedural frames of inference_bench_regime_<model>.npz, one image per call.  Pooled keep-set
    Jaccard against the reference's float32 lists, asserted against what the bf16-emulating oracle reaches
    (bf16_agreement.json "bench_regime"); scores on common rows within the bf16 score error."""
    from golden_util import BENCH_REGIME_OBJ_BIAS, bench_regime_frame
    g = np.load(os.path.join(GOLDEN, "inference_bench_regime_%s.npz" % model))
    M = MODES[mode]
    fixture = bf16_agreement(M["emulate"])["bench_regime"][model]
    dim = MODEL_DIMS[model]
    net = yolov3.Darknet(MODELS[model], device="cuda", dtype=M["dtype"])
    net.load_weights(golden_weights_path(model, obj_bias=BENCH_REGIME_OBJ_BIAS[model])).eval()
    for tag in ("a", "b"):
        pth, ith = g[tag + "_thresholds"]
        common = union = 0
        dps, cls_ok = [], []
        for name in (str(n) for n in g["names"]):
            frame = bench_regime_frame(name, dim)
            res = yolov3.inference(net, frame, device="cuda", prob_thresh=float(pth), nms_iou_thresh=float(ith), return_rows=True)[0]
            key = "%s_%s_" % (name, tag)
            rows, want = set(int(r) for r in res[3]), set(g[key + "rows"].tolist())
            common += len(rows & want)
            union += len(rows | want)
            gi = {int(r): k for k, r in enumerate(g[key + "rows"])}
            for k, r in enumerate(res[3]):
                if int(r) in gi:
                    dps.append(abs(float(res[1][k]) - float(g[key + "prob"][gi[int(r)]])))
                    cls_ok.append(int(res[2][k]) == int(g[key + "cls"][gi[int(r)]]))
        jac = common / max(union, 1)
        fl = fixture["all_" + tag]
        dps = np.array(dps) if dps else np.zeros(1)
        print("%s %s %s: pooled keep-set Jaccard %.3f (ideal %.3f; %d common of %d), score |d| median %.1e max %.1e, class agreement %.4f"
              % (mode, model, tag, jac, fl["jaccard"], common, union, np.median(dps), dps.max(), np.mean(cls_ok) if cls_ok else 1.0))
        if fl["union"] >= 100:
            assert jac >= fl["jaccard"] - 0.08
        else:
            assert (union - common) <= (fl["union"] - fl["common"]) + 4
        assert np.median(dps) <= M["regime_med"] and dps.max() <= M["regime_max"]
        assert not cls_ok or np.mean(cls_ok) >= 0.98


def test_bf16_keeps_planted_detections():
    """bf16 in a regime WITH margins.  The planted parameter set (tools/make_planted.py: procedural
    backbone, 19 x 19 head fitted so that each of the nine sample images has seven or eight confident detections -- kept scores
    >= 0.6, class margins ~1, every other box below 0.01) through the bf16 HIP path against the reference's float32
    ``inference()`` lists: the keep set must be the reference's (Jaccard >= 0.95 pooled, and within 0.02 of what the
    bf16-emulating oracle reaches: 1.0), classes identical on common rows, boxes within a few pixels (measured: 4).  Scores: the fitted
    head's objectness rows have norm ~200 and amplify the bf16 error of their input features forty-fold (tools/make_planted.py),
    so kept scores (>= 0.92 in float32) may move by up to 0.1 -- a pessimistic bound, asserted as such.  The float32 HIP path
    on the same parameters must reproduce the lists exactly."""
    from yolov3 import weights as W
    g = np.load(os.path.join(GOLDEN, "inference_planted_yolov3.npz"))
    for dtype in ("float32", "bf16", "fp16"):
        M = MODES.get(dtype, MODES["bf16"])
        floors = bf16_agreement(M["emulate"])["planted"]["yolov3"]
        net = yolov3.Darknet(MODELS["yolov3"], device="cuda", dtype=dtype).eval()
        net.set_params(W.planted_params(net.blocks, net.net_info))
        for tag in ("a", "b"):
            pth, ith = g[tag + "_thresholds"]
            common = union = same_cls = 0
            dps, dbox = [], []
            for name in (str(n) for n in g["names"]):
                frame = resize_bilinear_u8(load_jpeg_bgr("000000%s.jpg" % name), 608, 608)
                res = yolov3.inference(net, frame, device="cuda", prob_thresh=float(pth), nms_iou_thresh=float(ith), return_rows=True)[0]
                key = "%s_%s_" % (name, tag)
                gi = {int(r): k for k, r in enumerate(g[key + "rows"])}
                rows = [int(r) for r in res[3]]
                common += len(set(rows) & set(gi))
                union += len(set(rows) | set(gi))
                for k, r in enumerate(rows):
                    if r in gi:
                        same_cls += int(int(res[2][k]) == int(g[key + "cls"][gi[r]]))
                        dps.append(abs(float(res[1][k]) - float(g[key + "prob"][gi[r]])))
                        dbox.append(int(np.abs(res[0][k] - g[key + "tlbr"][gi[r]]).max()))
            jac = common / max(union, 1)
            fl = floors["all_" + tag]
            print("planted %s %s: keep-set Jaccard %.4f (ideal %.4f), %d common of %d, classes equal on %d, score |d| max %.2e, "
                  "box |d| max %d px" % (dtype, tag, jac, fl["jaccard"], common, union, same_cls, max(dps), max(dbox)))
            assert same_cls == common
            if dtype == "float32":
                assert jac == 1.0 and max(dps) <= 1e-3 and max(dbox) <= 1
            else:
                assert jac >= 0.95 and jac >= fl["jaccard"] - 0.02
                assert max(dps) <= M["planted_score"] and max(dbox) <= M["planted_box"]          # bf16: boxes of up to 400 px: 1.5 %
