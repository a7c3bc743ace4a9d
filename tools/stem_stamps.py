"""Phase cycles of the fused stem kernel (diagnostic library: make -C pytorch-yolov3_amd/csrc stamps; run with
Y3_HIP_LIB=pytorch-yolov3_amd/lib/libyolov3_hip_stamps.so)."""
import ctypes
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-yolov3_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import yolov3  # noqa: E402
from yolov3 import _hip, weights as W  # noqa: E402
from yolov3.cfgparse import parse_config  # noqa: E402
from yolov3.synthdata import synth_frames  # noqa: E402

cfg = os.path.join(ROOT, "pytorch-yolov3_amd", "models", "yolov3.cfg")
blocks, net_info = parse_config(cfg)
params = W.synth_params(blocks, net_info, seed=0, obj_bias=-8.5, calib=W.load_calibration("yolov3"))
net = yolov3.Darknet(cfg, device="cuda:0", dtype="bf16").eval()
net.set_params(params)
frames = torch.from_numpy(synth_frames(1, 16, 608, 608)).to("cuda:0")
lib = _hip.lib()
buf = (ctypes.c_ulonglong * 8)()
for it in range(3):
    net.forward_frames(frames, fresh=False)
    torch.cuda.synchronize()
    lib.y3_debug_stamps_fused(buf)
n = float(buf[7])
names = ["stem: fetch issue", "stem: image", "stem: patch store", "stem: barrier wait", "conv: taps", "conv: write-out",
         "conv: barrier wait"]
print("workgroups %d; cycles per workgroup (s_memtime, 100 MHz units x?):" % n)
for i in range(7):
    print("  %-20s %10.0f" % (names[i], buf[i] / n))
