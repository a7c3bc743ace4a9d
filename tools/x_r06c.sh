mkdir -p gpurun_out/r06c
L3="s76_128-256_k3,s38_256-512_k3,s19_512"
for v in "" e0 e0s e2; do
  if [ -n "$v" ]; then export Y3_HIP_LIB=$PWD/pytorch-yolov3_amd/lib/libyolov3_hip_$v.so; else unset Y3_HIP_LIB; fi
  python tools/conv_bench.py --only $L3 --variants igemm_v2,halo_ws_256,halo_ws,halo_dw > gpurun_out/r06c/cb3_${v:-default}.txt 2>&1
done
unset Y3_HIP_LIB
python tools/conv_bench.py --only s38_512-256_k1,s19_1024,s38_768,s76_256-128_k1 --variants igemm_v2,igemm_v3_ns3,dw_1x1,wres_1x1 > gpurun_out/r06c/cb1.txt 2>&1
python tools/conv_bench.py --dtype fp16 --only s38_512-256_k1,s19_1024,s38_256-512_k3 --variants igemm_v2,dw_1x1,halo_dw > gpurun_out/r06c/cb_f16.txt 2>&1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py -x -q -m gpu -k "direct_weights or private or kernel_choice or teacher or per_block or b16" > gpurun_out/r06c/parity_subset.log 2>&1
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --dump-ops gpurun_out/r06c/ops.txt > gpurun_out/r06c/bench.json 2> gpurun_out/r06c/bench.err
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --tuning auto_mask=16541 > gpurun_out/r06c/bench_no1x1dw.json 2> gpurun_out/r06c/bench_no1x1dw.err
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r06c/bench2.json 2> gpurun_out/r06c/bench2.err
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --tuning auto_mask=16541 > gpurun_out/r06c/bench_no1x1dw2.json 2> gpurun_out/r06c/bench_no1x1dw2.err
python bench.py --batch 1 --streams 1 --steps 50 --warmup 5 --no-cpu-baseline --no-extras --dump-ops gpurun_out/r06c/ops_b1.txt > gpurun_out/r06c/bench_b1.json 2> gpurun_out/r06c/bench_b1.err
python - > gpurun_out/r06c/profile_loop.txt 2>&1 <<'PY'
import cProfile, pstats, sys, os, time
sys.path.insert(0, "pytorch-yolov3_amd")
import numpy as np, torch, yolov3
from yolov3 import weights as W, stream as ystream
from yolov3.synthdata import synth_frames
cfg = "pytorch-yolov3_amd/models/yolov3.cfg"
net = yolov3.Darknet(cfg, device="cuda:0", dtype="bf16").eval()
net.set_params(W.synth_params(net.blocks, net.net_info, seed=0, obj_bias=-8.5, calib=W.load_calibration("yolov3")))
frames = synth_frames(555, 320, 608, 608)
list(ystream.detect_in_frames(net, (f for f in frames[:32]), batch_size=16))
t0 = time.perf_counter(); n = sum(1 for _ in ystream.detect_in_frames(net, (f for f in frames), batch_size=16)); print("fps", n / (time.perf_counter() - t0))
pr = cProfile.Profile(); pr.enable()
n = sum(1 for _ in ystream.detect_in_frames(net, (f for f in frames), batch_size=16))
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
PY
