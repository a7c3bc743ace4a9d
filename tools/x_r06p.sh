mkdir -p gpurun_out/r06p
python -m pytest tests -q -m gpu > gpurun_out/r06p/gpu_tests.log 2>&1
