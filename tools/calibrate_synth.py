"""One-off: measure per-BN-layer output statistics for procedural weights.

Walks each cfg with the REFERENCE's own modules (imported read-only through
tools/refshim.py), layer by layer: before a BN conv is evaluated its conv
output (pre-BN) mean/variance over a seeded batch is measured, rounded to
6 significant digits and installed (through yolov3.weights.synth_params) as
that layer's running statistics.  Result: pytorch-yolov3_amd/yolov3/
synth_calibration.json -- 2 scalars per BN layer, committed as data.
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from refshim import load_reference  # noqa: E402

ref = load_reference()
spec = importlib.util.spec_from_file_location(
    "amd_weights", os.path.join(HERE, "..", "pytorch-yolov3_amd", "yolov3", "weights.py"))
W = importlib.util.module_from_spec(spec)
spec.loader.exec_module(W)

CFG_DIR = os.path.join(HERE, "..", "pytorch-yolov3_amd", "models")
OUT = os.path.join(HERE, "..", "pytorch-yolov3_amd", "yolov3", "synth_calibration.json")


def calibrate(model, dim, seed=0):
    cfg = os.path.join(CFG_DIR, model + ".cfg")
    blocks, net_info = ref.darknet.parse_config(cfg)
    net = ref.Darknet(cfg, "cpu").eval()
    rs = np.random.RandomState(1234)
    frames = rs.randint(0, 256, size=(2, dim, dim, 3), dtype=np.uint8)
    x = torch.tensor(np.transpose(np.flip(frames, 3), (0, 3, 1, 2)).astype(np.float32) / 255.0)
    _, convs = W.conv_layout(blocks, net_info)
    conv_of_block = {c["block_idx"]: li for li, c in enumerate(convs)}
    calib = []
    cached = {}
    with torch.no_grad():
        for i, blk in enumerate(net.blocks):
            t = blk["type"]
            if t == "convolutional":
                li = conv_of_block[i]
                seq = net.modules_[i]
                conv = seq[0]
                if convs[li]["bn"]:
                    # weights of this conv do not depend on calib -> generate, measure
                    p = W.synth_params(blocks, net_info, seed=seed, calib=calib + [[0.0, 1.0]],
                                       upto_conv=li + 1)[li]
                    conv.weight.data.copy_(torch.from_numpy(p["weight"]))
                    y = conv(x)
                    m = float("%.6g" % float(y.mean()))
                    v = float("%.6g" % float(y.var(unbiased=False)))
                    calib.append([m, v])
                    p = W.synth_params(blocks, net_info, seed=seed, calib=calib, upto_conv=li + 1)[li]
                    bn = seq[1]
                    bn.weight.data.copy_(torch.from_numpy(p["bn_gamma"]))
                    bn.bias.data.copy_(torch.from_numpy(p["bn_beta"]))
                    bn.running_mean.copy_(torch.from_numpy(p["bn_mean"]))
                    bn.running_var.copy_(torch.from_numpy(p["bn_var"]))
                    x = seq[2](bn(y)) if len(seq) > 2 else bn(y)
                else:
                    p = W.synth_params(blocks, net_info, seed=seed, calib=calib, upto_conv=li + 1)[li]
                    conv.weight.data.copy_(torch.from_numpy(p["weight"]))
                    conv.bias.data.copy_(torch.from_numpy(p["bias"]))
                    x = seq(x)
            elif t in ("maxpool", "upsample"):
                x = net.modules_[i](x)
            elif t == "route":
                x = torch.cat([cached[j] for j in blk["layers"]], 1)
            elif t == "shortcut":
                x = cached[i - 1] + cached[i + blk["from"]]
            elif t == "yolo":
                pass
            if i in net.blocks_to_cache:
                cached[i] = x
            print(model, i, t, "rms %.3f" % float(x.pow(2).mean().sqrt()), flush=True)
    return calib


if __name__ == "__main__":
    table = {}
    for model, dim in (("yolov3-tiny", 416), ("yolov3", 608), ("yolov3-spp", 608)):
        table[model] = calibrate(model, dim)
    with open(OUT, "w") as fh:
        json.dump(table, fh, indent=0)
    print("wrote", OUT)
