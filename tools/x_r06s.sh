mkdir -p gpurun_out/r06s
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "small_grid" > gpurun_out/r06s/tests.log 2>&1
