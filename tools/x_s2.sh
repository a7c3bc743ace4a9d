#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s2
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "small_grid_direct_weights" > gpurun_out/s2/parity3.log 2>&1
tail -3 gpurun_out/s2/parity3.log
for b in 16 1 4; do
  echo "== batch $b"
  timeout 300 python tools/conv_bench.py --batch $b --only k3s2,s76_384-128_k1 --variants igemm_v2,igemm_v3_ns3,dw48_always --rounds 3 2>&1 | grep -v amdgpu.ids | cut -c1-330
done > gpurun_out/s2/s2_big.txt 2>&1
cat gpurun_out/s2/s2_big.txt
