#!/usr/bin/env python3
"""Event timeline of the PCIe-inclusive pipeline of bench.py (GPU box): per step, when its H2D copy and its forward start and end."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pytorch-yolov3_amd"))
import torch
import bench
args = bench.parse_args(["--no-cpu-baseline", "--no-extras"])
from yolov3 import _hip, weights as W
from yolov3.cfgparse import parse_config
dev = torch.device("cuda:0")
blocks, net_info = parse_config(os.path.join(ROOT, "pytorch-yolov3_amd", "models", "yolov3.cfg"))
params = W.synth_params(blocks, net_info, seed=0, obj_bias=-8.5, calib=W.load_calibration("yolov3"))
wl = bench.Workload("yolov3", 608, 16, "bf16", params, dev, 0, 1, 512, 3)
wl.enable_h2d(1)
h = wl.h2d
for i in range(12):
    wl.step(wl.frames, i, True)
torch.cuda.synchronize()
N = 18
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(N)]
base = torch.cuda.Event(enable_timing=True); base.record()
for i in range(N):
    k = i % wl.nstream; j = i % len(h["dev_frames"])
    with torch.cuda.stream(h["copy_stream"]):
        h["copy_stream"].wait_event(h["free_ev"][j])
        ev[i][0].record()
        if h["engine"] == "kernel":
            _hip.check(_hip.lib().y3_copy_bytes(h["host_frames"].data_ptr(), h["dev_frames"][j].data_ptr(), h["host_frames"].numel(), h["blocks"], _hip.stream_ptr()))
        else:
            h["dev_frames"][j].copy_(h["host_frames"], non_blocking=True)
        ev[i][1].record()
        h["ready_ev"][j].record(h["copy_stream"])
    with torch.cuda.stream(wl.streams[k]):
        wl.streams[k].wait_event(h["ready_ev"][j])
        ev[i][2].record()
        o = wl.net.forward_frames(h["dev_frames"][j], fresh=False, slot=k)
        h["free_ev"][j].record(wl.streams[k])
        wl.dets[k].run(o, wl.orig_hw, 0.05, 0.3)
        wl.gathers[k].run(wl.dets[k])
        ev[i][3].record()
torch.cuda.synchronize()
print("engine", h["engine"], "blocks", h["blocks"])
for i in range(N):
    t = [base.elapsed_time(e) for e in ev[i]]
    print("step %2d stream %d: copy %7.3f -> %7.3f (%.3f ms)   forward %7.3f -> %7.3f (%.3f ms)" % (i, i % 3, t[0], t[1], t[1] - t[0], t[2], t[3], t[3] - t[2]))
