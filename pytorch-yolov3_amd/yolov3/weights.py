"""Darknet ``.weights`` stream I/O and procedural (hash-generated) weights.

Stream layout consumed by the reference loader
(/root/reference/yolov3/darknet.py:415-476): 5 x int32 header, then float32
values; per ``[convolutional]`` block in cfg order either
``beta, gamma, running_mean, running_var`` (each Cout values, when the block
has a truthy ``batch_normalize``) or ``conv bias`` (Cout), followed by the conv
kernel in (Cout, Cin, kh, kw) order.

Real YOLOv3 checkpoints cannot be fetched here (no network), so tests and the
benchmark use *procedural* weights: every value is a pure function of
(seed, layer, tensor, element index) through a 64-bit integer mixer, which is
bit-reproducible on any machine (no libm, no RNG state).  The same generator
feeds the reference when the golden vectors are made (tools/make_goldens.py)
and this package on the GPU box, through the ordinary ``.weights`` file path.
"""
import math

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def conv_layout(blocks, net_info):
    """Per-block channel bookkeeping (reference darknet.py:229-313).

    Returns a list (one entry per block) of output channel counts, and a list
    of conv descriptors ``dict(block_idx, cin, cout, k, bn)`` in cfg order.
    """
    out_ch = []
    convs = []
    prev = net_info["channels"]
    cur = None
    for i, blk in enumerate(blocks):
        kind = blk["type"]
        if kind == "convolutional":
            cur = blk["filters"]
            convs.append(dict(
                block_idx=i, cin=prev, cout=cur, k=blk["size"],
                # builder keys on presence (darknet.py:237), loader on
                # presence AND truthiness (darknet.py:428)
                bn_module="batch_normalize" in blk,
                bn=bool(blk.get("batch_normalize", 0)),
            ))
        elif kind == "route":
            cur = sum(out_ch[j if j >= 0 else i + j] for j in blk["layers"])
        elif kind == "shortcut":
            if cur != out_ch[i + blk["from"]]:
                raise AssertionError(
                    "shortcut {} joins {} and {} channels".format(
                        i, cur, out_ch[i + blk["from"]]))
        # maxpool / upsample / yolo keep the channel count
        out_ch.append(cur)
        prev = cur
    return out_ch, convs


def stream_length(blocks, net_info):
    """Number of float32 values the loader consumes for this cfg."""
    _, convs = conv_layout(blocks, net_info)
    n = 0
    for c in convs:
        n += (4 if c["bn"] else 1) * c["cout"]
        n += c["cout"] * c["cin"] * c["k"] * c["k"]
    return n


def read_darknet_weights(path, blocks, net_info):
    """Read a ``.weights`` file -> (header int32[5], list of per-conv dicts).

    Each dict has ``weight`` (Cout,Cin,k,k) and either ``bn_beta, bn_gamma,
    bn_mean, bn_var`` or ``bias``.  A short file raises ``RuntimeError`` (the
    reference fails in ``view_as``); trailing bytes are ignored like the
    reference does.
    """
    with open(path, "rb") as fh:
        header = np.fromfile(fh, dtype=np.int32, count=5)
        stream = np.fromfile(fh, dtype=np.float32)
    _, convs = conv_layout(blocks, net_info)
    pos = 0

    def take(n, what, idx):
        nonlocal pos
        if pos + n > stream.size:
            raise RuntimeError(
                "weights file {!r} too short: block {} needs {} floats for {} "
                "at offset {}, file has {}".format(
                    path, idx, n, what, pos, stream.size))
        chunk = stream[pos:pos + n]
        pos += n
        return chunk

    params = []
    for c in convs:
        co, ci, k, bi = c["cout"], c["cin"], c["k"], c["block_idx"]
        entry = dict(block_idx=bi)
        if c["bn"]:
            entry["bn_beta"] = take(co, "bn beta", bi).copy()
            entry["bn_gamma"] = take(co, "bn gamma", bi).copy()
            entry["bn_mean"] = take(co, "bn mean", bi).copy()
            entry["bn_var"] = take(co, "bn var", bi).copy()
        else:
            entry["bias"] = take(co, "conv bias", bi).copy()
        entry["weight"] = take(co * ci * k * k, "conv weight", bi).reshape(
            co, ci, k, k).copy()
        params.append(entry)
    return header, params


def write_darknet_weights(path, params, header=None):
    """Write per-conv dicts (as returned by the reader) in Darknet order."""
    if header is None:
        header = np.array([0, 2, 0, 0, 0], dtype=np.int32)
    with open(path, "wb") as fh:
        np.asarray(header, dtype=np.int32).tofile(fh)
        for p in params:
            if "bn_beta" in p:
                for key in ("bn_beta", "bn_gamma", "bn_mean", "bn_var"):
                    np.ascontiguousarray(p[key], dtype=np.float32).tofile(fh)
            else:
                np.ascontiguousarray(p["bias"], dtype=np.float32).tofile(fh)
            np.ascontiguousarray(p["weight"], dtype=np.float32).tofile(fh)


# --------------------------------------------------------------------------
# procedural weights
# --------------------------------------------------------------------------

def hash_uniform(seed, stream, n):
    """n values k/2**24 (k integer in [0, 2**24)) as float64; pure integer mix.

    splitmix64-style finaliser over ``index * golden + key``; uint64 array
    arithmetic wraps modulo 2**64 in numpy, so the result is identical on any
    platform.
    """
    idx = np.arange(n, dtype=np.uint64)
    key = np.uint64(((int(seed) & 0xFFFFFFFF) << 32) ^ (int(stream) & 0xFFFFFFFF))
    with np.errstate(over="ignore"):
        x = idx * np.uint64(0x9E3779B97F4A7C15) + key * np.uint64(0xD1B54A32D192ED03)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return (x >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)


def _uniform(seed, stream, n, lo, hi):
    return (lo + (hi - lo) * hash_uniform(seed, stream, n)).astype(np.float32)


def load_calibration(cfg_name):
    """Per-BN-layer (mean, var) scalars measured once by tools/calibrate_synth.py.

    They stand in for trained running statistics: with them every conv output
    is roughly zero-mean / unit-variance before its LeakyReLU, so procedural
    weights give finite, O(1) activations through all 75 layers.
    """
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        "synth_calibration.json")
    with open(path, "r") as fh:
        table = json.load(fh)
    return table[cfg_name]


# per-attribute gain of the head (no-BN) conv rows: tx ty tw th obj, classes
_HEAD_GAIN_XY = 3.0
_HEAD_GAIN_WH = 2.0
_HEAD_GAIN_OBJ = 5.0
_HEAD_GAIN_CLS = 10.0


def synth_params(blocks, net_info, seed=0, obj_bias=-3.0, calib=None,
                 upto_conv=None):
    """Procedural parameters for every conv block of a cfg.

    BN convs: He-style uniform kernels for LeakyReLU(0.1); gamma ~ U(0.8,1.2)
    (halved on the conv that feeds a shortcut), beta ~ U(-0.1,0.1); running
    mean/var = the layer's calibrated scalars with a +-10 % per-channel jitter
    (``calib`` = list of (mean, var) per BN conv in cfg order, or None for
    (0, 1)).  Head convs (no BN): per-attribute gains so class scores are
    peaky, tw/th stay small, and ``obj_bias`` sets how many boxes pass the
    score threshold; a few classes get a positive bias so that per-class NMS
    has real overlaps to suppress.
    """
    _, convs = conv_layout(blocks, net_info)
    feeds_shortcut = set()
    for i, blk in enumerate(blocks):
        if blk["type"] == "shortcut":
            feeds_shortcut.add(i - 1)
    params = []
    bn_seen = 0
    for li, c in enumerate(convs):
        if upto_conv is not None and li >= upto_conv:
            break
        co, ci, k, bi = c["cout"], c["cin"], c["k"], c["block_idx"]
        fan_in = ci * k * k
        entry = dict(block_idx=bi)
        base = li * 8
        if c["bn"]:
            a = math.sqrt(3.0 * 2.0 / (1.01 * fan_in))
            entry["weight"] = _uniform(seed, base + 0, co * fan_in, -a, a).reshape(co, ci, k, k)
            g = 0.5 if bi in feeds_shortcut else 1.0
            entry["bn_gamma"] = _uniform(seed, base + 1, co, 0.8 * g, 1.2 * g)
            entry["bn_beta"] = _uniform(seed, base + 2, co, -0.1, 0.1)
            m_l, v_l = (0.0, 1.0) if calib is None or bn_seen >= len(calib) else calib[bn_seen]
            sd = math.sqrt(v_l)
            entry["bn_mean"] = (m_l + sd * (0.2 * hash_uniform(seed, base + 3, co) - 0.1)).astype(np.float32)
            entry["bn_var"] = (v_l * (0.9 + 0.2 * hash_uniform(seed, base + 4, co))).astype(np.float32)
            bn_seen += 1
        else:
            a = math.sqrt(3.0) / math.sqrt(fan_in)
            wgt = _uniform(seed, base + 0, co * fan_in, -a, a).reshape(co, fan_in)
            bias = _uniform(seed, base + 5, co, -0.5, 0.5)
            # yolo head channel = anchor * (5 + classes) + attr; 3 anchors/head
            if co % 3 == 0 and co // 3 > 5:
                n_attr = co // 3
                gain = np.full(n_attr, _HEAD_GAIN_CLS, dtype=np.float32)
                gain[0:2] = _HEAD_GAIN_XY
                gain[2:4] = _HEAD_GAIN_WH
                gain[4] = _HEAD_GAIN_OBJ
                wgt = wgt * np.tile(gain, 3)[:, None]
                for anc in range(3):
                    o = anc * n_attr
                    bias[o + 2] -= 0.5
                    bias[o + 3] -= 0.5
                    bias[o + 4] += obj_bias
                    bias[o + 5:o + n_attr:13] += 2.0   # a few popular classes
            entry["weight"] = wgt.astype(np.float32).reshape(co, ci, k, k)
            entry["bias"] = bias
        params.append(entry)
    return params


# --------------------------------------------------------------------------
# "planted" parameters: detections with margins (tools/make_planted.py)
# --------------------------------------------------------------------------

def install_planted_head(blocks, params, head_weight, head_bias):
    """Replace the FIRST detection head's conv (the block before the first [yolo]) of ``params`` by the given
    (Cout, Cin) weight / (Cout,) bias; returns ``params``."""
    first_yolo = next(i for i, b in enumerate(blocks) if b["type"] == "yolo")
    entry = next(p for p in params if p["block_idx"] == first_yolo - 1)
    if entry["weight"].shape[:2] != head_weight.shape or "bias" not in entry:
        raise ValueError("planted head does not fit block {}".format(first_yolo - 1))
    entry["weight"] = np.ascontiguousarray(head_weight, dtype=np.float32).reshape(entry["weight"].shape)
    entry["bias"] = np.ascontiguousarray(head_bias, dtype=np.float32)
    return params


def planted_params(blocks, net_info, model="yolov3", seed=0):
    """The procedural backbone with a FITTED 19 x 19 detection head (``planted_<model>.npz`` next to this file, made by
    tools/make_planted.py) and silent other heads: on the nine sample images (resized to the network's size) it returns a
    handful of confident, well-separated detections per image -- kept scores far from any threshold, class margins near
    one -- instead of the procedural heads' thousands of near-threshold boxes.  The fixture for "does bf16 keep real
    detections" (tests/golden/inference_planted_<model>.npz holds the reference's float32 lists)."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "planted_%s.npz" % model)
    with np.load(path) as z:
        head_w, head_b, silent = z["head_weight"], z["head_bias"], float(z["silent_obj_bias"])
    params = synth_params(blocks, net_info, seed=seed, obj_bias=silent, calib=load_calibration(model))
    return install_planted_head(blocks, params, head_w, head_b)
