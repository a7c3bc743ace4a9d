mkdir -p gpurun_out/r06_tests
python -m pytest tests -q -m gpu > gpurun_out/r06_tests/gpu_tests.log 2>&1
