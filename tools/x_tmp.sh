mkdir -p gpurun_out/r06z3
python -m pytest tests/test_gpu_parity.py tests/test_gpu_contention.py tests/test_gpu_bf16.py -x -q -m gpu -k "golden or kernel_choice or patch or halo or ws or mini or teacher or frames_are" > gpurun_out/r06z3/tests.log 2>&1
python tools/conv_bench.py --only s152_64-128_k3,s76_128-256_k3,s19_512-1024 --variants igemm_v2,halo_ws_256,patch_8x32 > gpurun_out/r06z3/cb.txt 2>&1
python tools/conv_bench.py --dtype float32 --batch 8 --only s76_128-256_k3,s19_512-1024,s38_256 --variants igemm_v2,halo_ws_256 > gpurun_out/r06z3/cb_f32.txt 2>&1
