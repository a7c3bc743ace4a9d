"""Host cost of one bench step: wall time the Python thread spends enqueueing a step, CPU seconds of the whole process
per step (runtime threads included), with and without hipGraph replay of the forward plan.  Decides whether 8 ranks fit
the GPU box's host cores (the cgroup grants 16 hardware threads)."""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-yolov3_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from yolov3 import _hip, weights as W  # noqa: E402
from yolov3.cfgparse import parse_config  # noqa: E402

blocks, net_info = parse_config(os.path.join(ROOT, "pytorch-yolov3_amd", "models", "yolov3.cfg"))
params = W.synth_params(blocks, net_info, seed=0, obj_bias=-8.5, calib=W.load_calibration("yolov3"))
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
print("host cores granted: %d" % len(os.sched_getaffinity(0)))
for graph in (0, 1, 0, 1):
    opts = {"auto_mask": _hip.options().auto_mask | _hip.AM_HALO_TILE256, "use_graph": graph}
    wl = bench.Workload("yolov3", 608, 16, "bf16", params, dev, 0, 1, 512, 3, options=opts)
    for i in range(12):
        wl.step(wl.frames, i)
    torch.cuda.synchronize()
    n = 100
    c0 = time.process_time()
    t0 = time.perf_counter()
    for i in range(n):
        wl.step(wl.frames, i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    c1 = time.process_time()
    print("use_graph=%d: enqueue %.3f ms/step (host thread), %.3f ms/step wall, %.3f CPU-ms/step (all threads), %.0f frames/s"
          % (graph, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, (c1 - c0) / n * 1e3, 16 * n / (t2 - t0)))
    del wl
