"""Time the fused detection tail (y3_detect) alone on the forward outputs of the bench workload."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-yolov3_amd"))
import yolov3  # noqa: E402
from yolov3 import weights as W  # noqa: E402
from yolov3.inference import Detector  # noqa: E402
from yolov3.synthdata import synth_frames  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--obj-bias", type=float, nargs="+", default=[-8.5, -5.0, -3.0])
    ap.add_argument("--batch", type=int, default=16)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = os.path.join(ROOT, "pytorch-yolov3_amd", "models", "yolov3.cfg")
    for ob in args.obj_bias:
        net = yolov3.Darknet(cfg, device="cuda:0", dtype="bf16").eval()
        net.set_params(W.synth_params(net.blocks, net.net_info, seed=0, obj_bias=ob, calib=W.load_calibration("yolov3")))
        frames = torch.from_numpy(synth_frames(123, args.batch, 608, 608)).to(dev)
        out = net.forward_frames(frames, fresh=False)
        rows = out["class_prob"].shape[1]
        det = Detector(args.batch, rows, dev)
        hw = torch.tensor([[608, 608]] * args.batch, dtype=torch.int32, device=dev)
        for _ in range(3):
            det.run(out, hw, 0.05, 0.3)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            det.run(out, hw, 0.05, 0.3)
        e1.record()
        torch.cuda.synchronize()
        cand = int((out["class_prob"] >= 0.05).sum()) / args.batch
        kept = float(det.count.float().mean())
        print("obj_bias %.1f: %.1f candidates/frame, %.1f kept/frame, detect %.4f ms per batch of %d" % (
            ob, cand, kept, e0.elapsed_time(e1) / 50, args.batch))


if __name__ == "__main__":
    main()
