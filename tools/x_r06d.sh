mkdir -p gpurun_out/r06d
L3="s76_128-256_k3,s38_256-512_k3,s19_512"
for v in "" pf1 "" pf1; do
  if [ -n "$v" ]; then export Y3_HIP_LIB=$PWD/pytorch-yolov3_amd/lib/libyolov3_hip_$v.so; else unset Y3_HIP_LIB; fi
  python tools/conv_bench.py --only $L3 --variants igemm_v2,halo_ws_256,halo_dw >> gpurun_out/r06d/cb3_${v:-default}.txt 2>&1
done
unset Y3_HIP_LIB
python -m pytest tests/test_gpu_parity.py tests/test_gpu_contention.py -x -q -m gpu -k "direct_weights or private or kernel_choice or counted_wait" > gpurun_out/r06d/parity_subset.log 2>&1
for i in 1 2; do
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r06d/bench_$i.json 2> gpurun_out/r06d/bench_$i.err
Y3_HIP_LIB=$PWD/pytorch-yolov3_amd/lib/libyolov3_hip_pf1.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r06d/bench_pf1_$i.json 2> gpurun_out/r06d/bench_pf1_$i.err
done
