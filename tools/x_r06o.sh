mkdir -p gpurun_out/r06o
python -m pytest tests/test_gpu_parity.py tests/test_gpu_contention.py tests/test_gpu_properties.py tests/test_gpu_pipeline.py tests/test_gpu_bf16.py -x -q -m gpu > gpurun_out/r06o/tests.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06o/bench.json 2> gpurun_out/r06o/bench.err
