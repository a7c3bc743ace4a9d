import sys, os, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"), "pytorch-yolov3_amd")); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","/root/repo"))
import torch, bench
from yolov3 import weights as W
from yolov3.cfgparse import parse_config
ROOT=bench.ROOT
blocks, net_info = parse_config(os.path.join(ROOT, "pytorch-yolov3_amd", "models", "yolov3.cfg"))
params = W.synth_params(blocks, net_info, seed=0, obj_bias=-8.5, calib=W.load_calibration("yolov3"))
dev=torch.device("cuda",0); torch.cuda.set_device(0)
wl=bench.Workload("yolov3",608,16,"bf16",params,dev,0,1,512,3)
for i in range(10): wl.step(wl.frames,i)
torch.cuda.synchronize()
t0=time.perf_counter()
for i in range(60): wl.step(wl.frames,i)
t1=time.perf_counter()
torch.cuda.synchronize()
t2=time.perf_counter()
print("host enqueue per step %.3f ms; total per step %.3f ms" % ((t1-t0)/60*1e3,(t2-t0)/60*1e3))
