"""CPU ORACLE for the YOLOv3 hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A CPU restatement (torch-CPU fp32 ops + numpy) of the reference's
``Darknet.forward`` -> YOLO decode -> threshold -> int/tlbr -> per-class NMS
path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module, and only as the checker / the
timed CPU baseline; the product package (pytorch-yolov3_amd/yolov3) never
imports it and raises when the HIP library is missing.

PARITY PINNED: every function here is checked in ``tests/test_oracle_golden.py``
(``-m "not gpu"``) against golden vectors produced by importing the real
reference in the build container (tools/make_goldens.py -> tests/golden/*.npz),
and against the reference's own known-answer vector for ``cxywh_to_tlbr``
(/root/reference/tests/test_inference.py:12-23).

The arithmetic of the reference lives in third-party libraries that are not
vendored (torch: Conv2d/BatchNorm2d/LeakyReLU/max_pool2d/Upsample/softmax;
numpy: argsort and int64/float64 array arithmetic; both unpinned in
/root/reference/requirements.txt).  This restatement calls the same public
torch/numpy primitives, op by op and in the reference's order, so it is the
"-d cpu" path with the nn.Module plumbing removed.

bf16 EMULATION (``OracleDarknet.forward(..., emulate_bf16=True)``): the reference
is float32 only, so the bf16 throughput mode of the product has no reference
counterpart.  Its checker is THIS pinned float32 restatement with bf16 storage
rounding (round-to-nearest-even, ``torch.bfloat16``) inserted exactly where the
bf16 plan keeps a tensor in HBM: the normalised input, every conv weight, every
conv / shortcut output that is materialised.  Everything between two rounding
points is the reference's float32 op sequence (conv accumulate, BatchNorm,
LeakyReLU, the shortcut add: /root/reference/yolov3/darknet.py:244-257,376-379);
detection-head convs stay float32 (their logits are decoded in float32).  The
rounding points are listed in ``bf16_rounding_points``.  ``accumulate="f64"``
sums every conv in float64 (bf16 x bf16 products are exact there): the distance
between the f32- and f64-accumulating runs is the summation-order noise two
correct bf16 implementations may differ by, which is what the GPU tests derive
their tolerances from (tests/test_oracle_golden.py::test_bf16_emulation_noise_floor).
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

try:
    from . import ref_io              # oracle-side cfg / .weights readers, independent of the product's
except ImportError:                   # imported as a top-level module
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import ref_io

BN_EPS = 1e-5          # torch.nn.BatchNorm2d default, reference darknet.py:252
LEAKY_SLOPE = 0.1      # reference darknet.py:256


# --------------------------------------------------------------------------
# per-op restatements
# --------------------------------------------------------------------------

def bf16_round(t):
    """float32 tensor -> nearest bfloat16 (ties to even) -> float32: the storage rounding of the bf16 mode."""
    return t.to(torch.bfloat16).to(torch.float32)


def f16_round(t):
    """float32 tensor -> nearest IEEE half (ties to even, subnormals kept, overflow -> inf) -> float32: the storage
    rounding of the fp16 mode."""
    return t.to(torch.float16).to(torch.float32)


STORAGE_ROUND = {"bf16": bf16_round, "f16": f16_round}


def storage_round(emulate):
    """``emulate``: None / False (float32, the reference), True or "bf16", "f16" -> the rounding function (or None)."""
    if not emulate:
        return None
    return STORAGE_ROUND["bf16" if emulate is True else str(emulate)]


def _conv2d(x, w, stride, pad, accumulate):
    if accumulate == "f64":
        return F.conv2d(x.double(), w.double(), None, stride=stride, padding=pad).float()
    return F.conv2d(x, w, None, stride=stride, padding=pad)


def conv_block(x, p, stride, pad, leaky, bf16_weights=False, accumulate="f32", round_weights=None):
    """conv -> [BN eval] -> [LeakyReLU 0.1]  (reference darknet.py:236-264, run :367-368).

    x: (B,Cin,H,W) f32.  p: dict with ``weight`` and either BN tensors or
    ``bias`` (numpy).  ``activation=linear`` means identity (darknet.py:258-261).
    ``bf16_weights`` / ``round_weights`` ("bf16" / "f16"): kernel weights rounded to that storage type first
    (16-bit emulation; BN tensors and biases stay float32).  The result is float32 and NOT rounded here.
    """
    w = torch.from_numpy(np.ascontiguousarray(p["weight"]))
    rnd = storage_round(round_weights if round_weights else bf16_weights)
    if rnd is not None:
        w = rnd(w)
    if "bn_gamma" in p:
        y = _conv2d(x, w, stride, pad, accumulate)
        y = F.batch_norm(
            y, torch.from_numpy(p["bn_mean"].copy()), torch.from_numpy(p["bn_var"].copy()),
            torch.from_numpy(p["bn_gamma"].copy()), torch.from_numpy(p["bn_beta"].copy()),
            training=False, eps=BN_EPS)
    elif accumulate == "f64":
        y = (F.conv2d(x.double(), w.double(), torch.from_numpy(p["bias"].copy()).double(), stride=stride,
                      padding=pad)).float()
    else:
        y = F.conv2d(x, w, torch.from_numpy(p["bias"].copy()), stride=stride, padding=pad)
    if leaky:
        y = F.leaky_relu(y, LEAKY_SLOPE)
    return y


def maxpool(x, size, stride):
    """Reference MaxPool2d.forward (darknet.py:16-29).

    size>1 and stride==1: ZERO-pad right/bottom by size-1, then pool with no
    padding (windows extend down-right, out-of-bounds = 0.0, not -inf).
    """
    if size > 1 and stride == 1:
        x = F.pad(x, (0, size - 1, 0, size - 1), mode="constant", value=0.0)
    return F.max_pool2d(x, size, stride, 0)


def upsample(x, factor):
    """nn.Upsample(scale_factor, mode='nearest') (darknet.py:299-305)."""
    return F.interpolate(x, scale_factor=factor, mode="nearest")


def yolo_decode(x, anchors):
    """YOLOLayer.forward (darknet.py:48-122) for one head.

    x: (B, A*(5+C), h, w).  anchors: list of A (w, h) pixel pairs (already
    selected by ``mask``).  Returns bbox_xywh (B,A*h*w,4) f32 with x,y in
    grid-normalised units and w,h in PIXELS (the /net size happens in
    ``forward``), class_prob (B,A*h*w) f32, class_idx (B,A*h*w) i64.
    Row order a*h*w + y*w + x.
    """
    b, ch, h, w = x.shape
    na = len(anchors)
    nattr = ch // na
    t = x.reshape(b, na, nattr, h, w)
    gx = torch.arange(w, dtype=torch.float32).reshape(1, 1, 1, w)
    gy = torch.arange(h, dtype=torch.float32).reshape(1, 1, h, 1)
    aw = torch.tensor([a[0] for a in anchors], dtype=torch.float32).reshape(1, na, 1, 1)
    ah = torch.tensor([a[1] for a in anchors], dtype=torch.float32).reshape(1, na, 1, 1)
    bx = (torch.sigmoid(t[:, :, 0]) + gx) / w
    by = (torch.sigmoid(t[:, :, 1]) + gy) / h
    bw = torch.exp(t[:, :, 2]) * aw
    bh = torch.exp(t[:, :, 3]) * ah
    obj = torch.sigmoid(t[:, :, 4])
    cls = torch.softmax(t[:, :, 5:], dim=2)
    best, idx = torch.max(cls, dim=2)
    prob = best * obj
    bbox = torch.stack((bx, by, bw, bh), dim=-1).reshape(b, na * h * w, 4)
    return bbox, prob.reshape(b, -1), idx.reshape(b, -1)


# --------------------------------------------------------------------------
# whole-network forward
# --------------------------------------------------------------------------

class OracleDarknet:
    """Functional restatement of reference ``Darknet`` (darknet.py:318-476)."""

    def __init__(self, config_fpath):
        self.blocks, self.net_info = ref_io.read_cfg(config_fpath)
        # negative route indices -> absolute (darknet.py:338-343)
        for i, blk in enumerate(self.blocks):
            if blk["type"] == "route":
                blk["layers"] = [j if j >= 0 else i + j for j in blk["layers"]]
        conv_blocks = [i for i, blk in enumerate(self.blocks) if blk["type"] == "convolutional"]
        self._conv_slot = {bi: n for n, bi in enumerate(conv_blocks)}
        self.params = None
        self.header = None

    def load_weights(self, path):
        self.header, self.params = ref_io.read_weights(path, self.blocks, self.net_info["channels"])
        return self

    def set_params(self, params):
        self.params = params
        return self

    def bf16_rounding_points(self):
        """Which block outputs the bf16 mode stores in bf16 (True), keeps in float32 (False: detection-head convs,
        whose logits go to the decode in float32, and convs whose ONLY reader is the shortcut right after them:
        the product's conv epilogue adds the shortcut operand in float32 and stores the sum once)."""
        n = len(self.blocks)
        readers = [0] * n
        for i, blk in enumerate(self.blocks):
            if blk["type"] in ("convolutional", "maxpool", "upsample", "yolo") and i > 0:
                readers[i - 1] += 1
            elif blk["type"] == "route":
                for j in blk["layers"]:
                    readers[j] += 1
            elif blk["type"] == "shortcut":
                readers[i - 1] += 1
                readers[i + blk["from"]] += 1
        rounds = [True] * n
        for i, blk in enumerate(self.blocks):
            if blk["type"] != "convolutional":
                continue
            nxt = self.blocks[i + 1]["type"] if i + 1 < n else None
            if nxt == "yolo":
                rounds[i] = False
            elif nxt == "shortcut" and readers[i] == 1 and i + 1 + self.blocks[i + 1]["from"] != i:
                rounds[i] = False
        return rounds

    def forward(self, x, collect=None, emulate_bf16=False, accumulate="f32", emulate=None):
        """x: torch (B,3,H,W) f32.  Returns dict like the reference (darknet.py:401-405).

        ``collect``: optional dict filled with {block_idx: tensor} of every
        block output (used by per-layer parity tests).
        ``emulate_bf16`` / ``accumulate``: see the module docstring (checker of the bf16 mode).
        ``emulate``: None, "bf16" (== ``emulate_bf16=True``) or "f16": the same rounding points with IEEE-half
        storage (the product's ``dtype="fp16"`` mode: same kernels, ``v_mfma_f32_16x16x32_f16``).
        """
        outs = []
        heads = []
        emulate = emulate or ("bf16" if emulate_bf16 else None)
        rnd = storage_round(emulate)
        rounds = self.bf16_rounding_points() if rnd is not None else None
        with torch.no_grad():
            if rnd is not None:
                x = rnd(x)
            for i, blk in enumerate(self.blocks):
                kind = blk["type"]
                if kind == "convolutional":
                    k = blk["size"]
                    pad = (k - 1) // 2 if "pad" in blk else 0      # darknet.py:240
                    x = conv_block(x, self.params[self._conv_slot[i]], blk["stride"], pad,
                                   blk["activation"] == "leaky", round_weights=emulate, accumulate=accumulate)
                    if rnd is not None and rounds[i]:
                        x = rnd(x)
                elif kind == "maxpool":
                    x = maxpool(x, blk["size"], blk["stride"])
                elif kind == "upsample":
                    x = upsample(x, blk["stride"])
                elif kind == "route":
                    x = torch.cat([outs[j] for j in blk["layers"]], dim=1)   # darknet.py:372-375
                elif kind == "shortcut":
                    x = outs[i - 1] + outs[i + blk["from"]]                  # darknet.py:379
                    if rnd is not None:
                        x = rnd(x)
                elif kind == "yolo":
                    anchors = [blk["anchors"][m] for m in blk["mask"]]       # darknet.py:44
                    heads.append(yolo_decode(x, anchors))
                outs.append(x)
                if collect is not None:
                    collect[i] = x
            bbox = torch.cat([h[0] for h in heads], dim=1)
            prob = torch.cat([h[1] for h in heads], dim=1)
            idx = torch.cat([h[2] for h in heads], dim=1)
            # w,h divided by the CFG's net size, whatever the input size (darknet.py:395-399)
            bbox[:, :, 2] = bbox[:, :, 2] / self.net_info["width"]
            bbox[:, :, 3] = bbox[:, :, 3] / self.net_info["height"]
        return {"bbox_xywh": bbox, "class_prob": prob, "class_idx": idx}


# --------------------------------------------------------------------------
# post-processing (numpy)
# --------------------------------------------------------------------------

def frames_to_input(frames):
    """uint8 BGR HxWx3 frames -> (B,3,H,W) f32 RGB/255 (reference inference.py:332-333)."""
    arr = np.stack(frames)[:, :, :, ::-1]
    return np.ascontiguousarray(np.transpose(arr, (0, 3, 1, 2))).astype(np.float32) / 255.0


def cxywh_to_tlbr(box):
    """Integer centre/size -> corners, floor-halving (reference inference.py:269-283)."""
    out = np.array(box, copy=True)
    half = box[:, 2:4] // 2
    out[:, 0:2] = box[:, 0:2] - half
    out[:, 2:4] = box[:, 0:2] + half
    return out


def nms_single(tlbr, prob, iou_thresh=0.3):
    """Greedy NMS over one set of boxes (reference inference.py:161-217).

    +1 pixel widths, IoU in float64 (int64 / int64 true division), suppress
    iff iou > thresh (strict).  Visit order: ``np.argsort(prob)[::-1]``.
    Returns indices in pick order (numpy int64 scalars like the reference).
    """
    tlbr = np.asarray(tlbr)
    n = tlbr.shape[0]
    if n == 0:
        return []
    x1, y1, x2, y2 = tlbr[:, 0], tlbr[:, 1], tlbr[:, 2], tlbr[:, 3]
    area = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = np.argsort(prob)[::-1]
    alive = np.ones(n, dtype=bool)
    keep = []
    for pos in range(n):
        i = order[pos]
        if not alive[i]:
            continue
        keep.append(i)
        rest = order[pos + 1:]
        rest = rest[alive[rest]]
        if rest.size == 0:
            continue
        iw = np.maximum(0, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + 1)
        ih = np.maximum(0, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + 1)
        inter = iw * ih
        iou = inter / (area[i] + area[rest] - inter)
        alive[rest[iou > iou_thresh]] = False
    return keep


def non_max_suppression(tlbr, class_prob, class_idx=None, iou_thresh=0.3):
    """Per-class (or class-agnostic) NMS (reference inference.py:220-266)."""
    if class_idx is None:
        return nms_single(tlbr, class_prob, iou_thresh)
    keep = []
    for cls in set(class_idx):                      # same iteration order as the reference
        members = np.where(class_idx == cls)[0]
        sub = nms_single(tlbr[members], class_prob[members], iou_thresh)
        keep.extend(members[sub].tolist())
    return keep


def postprocess(bbox_xywh, class_prob, class_idx, orig_shapes, prob_thresh=0.05,
                nms_iou_thresh=0.3, audit=False):
    """Tail of reference ``inference()`` (inference.py:338-366) on numpy arrays.

    orig_shapes: per-frame (H, W[, C]).  Returns per frame
    [tlbr int64 (K,4), prob f32 (K,), cls int64 (K,)].

    ``audit=True`` appends, per frame, the prediction row of every kept detection, the rows of all
    candidates (score >= threshold) and which candidates are FRAGILE: a scaled coordinate within 2e-3 px
    of an integer or a score within 1e-5 of the threshold, i.e. where a float difference of a few ulp
    between two correct implementations legitimately flips ``astype(int)`` / the threshold test (same
    audit as tools/make_goldens.py stores with the G7 goldens).
    """
    results = []
    mask = class_prob >= prob_thresh
    for i in range(bbox_xywh.shape[0]):
        box = bbox_xywh[i, mask[i], :].copy()
        prob = class_prob[i, mask[i]]
        cls = class_idx[i, mask[i]]
        box[:, [0, 2]] *= orig_shapes[i][1]
        box[:, [1, 3]] *= orig_shapes[i][0]
        with np.errstate(invalid="ignore"):
            tlbr = cxywh_to_tlbr(box.astype(np.int64))
        keep = non_max_suppression(tlbr, prob, class_idx=cls, iou_thresh=nms_iou_thresh)
        item = [tlbr[keep, :], prob[keep], cls[keep]]
        if audit:
            cand = np.where(mask[i])[0]
            with np.errstate(invalid="ignore"):
                dist = np.abs(box - np.rint(box)).min(axis=1) if len(cand) else np.zeros(0)
            fragile = (dist < 2e-3) | (np.abs(prob - np.float32(prob_thresh)) < 1e-5)
            item += [cand[keep].astype(np.int64), cand.astype(np.int64), fragile]
        results.append(item)
    return results


def compare_detections(got, want):
    """One frame of the product's detections, ``got`` = [tlbr, prob, cls, rows], against an audited oracle
    frame (``postprocess(..., audit=True)``): the kept prediction rows must be the same set, classes equal,
    boxes equal, except at FRAGILE candidates (see ``postprocess``), where the truncated pixel may differ by
    one.  Returns (rows that differ in the keep set, fragile boxes that differ); raises AssertionError on
    any difference outside the fragile rows."""
    tlbr, prob, cls, rows = got[0], got[1], got[2], got[3]
    w_tlbr, w_prob, w_cls, w_rows, w_cand, w_frag = want
    fragile_rows = set(w_cand[w_frag].tolist())
    g = {int(r): k for k, r in enumerate(rows)}
    w = {int(r): k for k, r in enumerate(w_rows)}
    assert len(g) == len(rows), "duplicate rows in detections"
    diff = set(g) ^ set(w)
    assert diff <= fragile_rows, "keep sets differ at non-fragile rows %s" % sorted(diff - fragile_rows)[:8]
    bad = 0
    for r in set(g) & set(w):
        a, b = g[r], w[r]
        assert cls[a] == w_cls[b], "class differs at row %d" % r
        if not (tlbr[a] == w_tlbr[b]).all():
            assert r in fragile_rows and np.abs(tlbr[a] - w_tlbr[b]).max() <= 1, \
                "box differs at non-fragile row %d: %s vs %s" % (r, tlbr[a], w_tlbr[b])
            bad += 1
    return len(diff), bad


def inference(net, frames, prob_thresh=0.05, nms_iou_thresh=0.3):
    """Net-sized uint8 BGR frames -> detections, like reference ``inference()``."""
    if not isinstance(frames, list):
        frames = [frames]
    out = net.forward(torch.from_numpy(frames_to_input(frames)))
    return postprocess(out["bbox_xywh"].numpy(), out["class_prob"].numpy(),
                       out["class_idx"].numpy(), [f.shape for f in frames],
                       prob_thresh, nms_iou_thresh)


def canonical_rows(det):
    """Sort one frame's [tlbr, prob, cls] by (cls, -prob, box) for set-style comparison."""
    tlbr, prob, cls = det
    if len(prob) == 0:
        return tlbr, prob, cls
    key = np.lexsort((tlbr[:, 3], tlbr[:, 2], tlbr[:, 1], tlbr[:, 0], -prob.astype(np.float64), cls))
    return tlbr[key], prob[key], cls[key]
