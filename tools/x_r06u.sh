mkdir -p gpurun_out/r06u
python -m pytest tests/test_gpu_a_fresh_process.py tests/test_callers.py -x -q -m gpu > gpurun_out/r06u/tests.log 2>&1
Y3_BENCH_FORCE_LAUNCH=1 python bench.py --gpus 1 --steps 10 --warmup 3 > gpurun_out/r06u/bench_launch1.json 2> gpurun_out/r06u/bench_launch1.err
