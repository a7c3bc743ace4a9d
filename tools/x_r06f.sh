mkdir -p gpurun_out/r06f
python tools/conv_bench.py --only s19_512-256_k1,s38_512-256_k1 --variants igemm_v2,igemm_v3_ns3,dw_1x1 > gpurun_out/r06f/cb1.txt 2>&1
python tools/conv_bench.py --batch 8 --only s19_512-256_k1,s38_512-256_k1,s19_1024 --variants igemm_v2,igemm_v3_ns3,dw_1x1 > gpurun_out/r06f/cb1_b8.txt 2>&1
python tools/conv_bench.py --batch 4 --only s38_512-256_k1,s19_1024 --variants igemm_v2,igemm_v3_ns3,dw_1x1 > gpurun_out/r06f/cb1_b4.txt 2>&1
python tools/detect_bench.py --obj-bias -8.5 -6.9 -5.0 > gpurun_out/r06f/detect_bench.txt 2>&1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_contention.py tests/test_gpu_pipeline.py -x -q -m gpu -k "nms or inference or detect or dw1x1 or direct_weights_1x1 or pipeline" > gpurun_out/r06f/tests.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency > gpurun_out/r06f/bench.json 2> gpurun_out/r06f/bench.err
