# Same-box A/B of lib/libyolov3_hip_old.so (a build of an earlier tree) against the current library: parity subset, per-layer
# conv_bench, clock stamps, end-to-end bench.py (three interleaved pairs).
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py -x -q 2>&1 | grep -E "^(FAILED|ERROR|E  )|passed|failed" | head -20
bash tools/ab_lib.sh pytorch-yolov3_amd/lib/libyolov3_hip_old.so pytorch-yolov3_amd/lib/libyolov3_hip.so --only s76_128-256_k3,s38_256-512_k3,s19_512 --variants halo_ws_256 --rounds 5
Y3_HIP_LIB=pytorch-yolov3_amd/lib/libyolov3_hip_stamps.so python tools/conv_bench.py --only s76_128-256_k3,s38_256-512_k3,s19_512 --variants halo_ws_256 --stamps 2>&1 | grep -v amdgpu
for rep in 1 2 3; do for L in old new; do
  if [ $L = old ]; then export Y3_HIP_LIB=pytorch-yolov3_amd/lib/libyolov3_hip_old.so; else unset Y3_HIP_LIB; fi
  timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 80 --warmup 10 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$L value %8.1f  ms/step %.4f  dominant %7.1f TF' % (d['value'], d['ms_per_step'], d['roofline']['achieved']))
"
done; done
