#!/usr/bin/env python3
"""BASELINE.md section 4 sanity anchor (runs in the BUILD container only, where /root/reference exists): the CPU
restatement that bench.py times on the GPU node (oracle/) against the true reference's ``inference()`` on the same
machine, same procedural weights and frames.  The restatement's GPU-node number is trusted only if the two agree
within ~10 % here.  Usage: python tools/ref_vs_oracle_timing.py [threads]"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_goldens as MG  # noqa: E402  (imports the reference as ``yolov3`` and OUR modules under another name)

from oracle import darknet_oracle as orc  # noqa: E402  (imports nothing of the product)

ref, W, SD = MG.ref, MG.W, MG.SD


def median_s(fn, n):
    fn()
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
    torch.set_num_threads(threads)
    print("torch %s, %d threads" % (torch.__version__, threads))
    for model, dim in (("yolov3", 608), ("yolov3-tiny", 416)):
        cfg = MG.MODELS[model]["cfg"]
        blocks, net_info = ref.darknet.parse_config(cfg)
        params = W.synth_params(blocks, net_info, seed=0, obj_bias=-8.5, calib=W.load_calibration(model))
        wpath = "/tmp/ref_vs_oracle_%s.weights" % model
        W.write_darknet_weights(wpath, params)
        rnet = ref.Darknet(cfg, device="cpu")
        rnet.load_weights(wpath)
        rnet.eval()
        onet = orc.OracleDarknet(cfg).load_weights(wpath)
        frames = [f for f in SD.synth_frames(123, 1, dim, dim)]
        t_ref = median_s(lambda: ref.inference(rnet, frames, device="cpu", prob_thresh=0.05, nms_iou_thresh=0.3), 7)
        t_orc = median_s(lambda: orc.inference(onet, frames, 0.05, 0.3), 7)
        print("%-12s %d^2 batch 1, inference(): reference %7.1f ms = %5.2f frames/s | restatement %7.1f ms = %5.2f frames/s | "
              "restatement / reference time %.3f" % (model, dim, t_ref * 1e3, 1 / t_ref, t_orc * 1e3, 1 / t_orc, t_orc / t_ref))


if __name__ == "__main__":
    main()
