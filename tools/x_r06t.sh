mkdir -p gpurun_out/r06t
python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_gpu_properties.py -x -q -m gpu -k "nms or inference or detect or pipeline or golden or exact or bench_regime or crops" > gpurun_out/r06t/tests.log 2>&1
python tools/detect_bench.py --obj-bias -8.5 -6.9 -5.0 -3.0 > gpurun_out/r06t/detect_bench.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency > gpurun_out/r06t/bench.json 2> gpurun_out/r06t/bench.err
