#!/bin/bash
# experiment: upper bound of what hiding the halo kernels' epilogue (persistent form: prologue already hidden) is worth
L=pytorch-yolov3_amd/lib
CB="python tools/conv_bench.py --only s76_128-256_k3,s38_256-512_k3,s19_512,s152_64 --variants halo_ws,halo_wsp,patch_8x32,halo_ws_256"
echo "== conv_bench, product library"; $CB 2>&1 | grep -v amdgpu | cut -c1-330
echo "== conv_bench, no-epilogue persistent kernels"; Y3_HIP_LIB=$L/libyolov3_hip_noepi.so Y3_CONV_BENCH_DEBUG=1 $CB 2>&1 | grep -v amdgpu | cut -c1-330
B="python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 10"
show() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], {k:(v['ms'],v['launches']) for k,v in d['kernels'].items() if 'halo' in k or 'patch' in k})
"; }
echo "== e2e default"; $B 2>/dev/null | show
echo "== e2e persistent"; $B --tuning halo_persistent=1 2>/dev/null | show
echo "== e2e persistent, no epilogue in the timed region"; Y3_HIP_LIB=$L/libyolov3_hip_noepi.so Y3_BENCH_DEBUG_AFTER_WARMUP=1 $B --tuning halo_persistent=1 2>/dev/null | show
echo "== e2e default again"; $B 2>/dev/null | show
