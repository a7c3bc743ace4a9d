mkdir -p gpurun_out/r06w
python -m pytest tests/test_gpu_a_fresh_process.py tests/test_gpu_pipeline.py tests/test_callers.py -x -q -m gpu > gpurun_out/r06w/tests.log 2>&1
for i in 1 2; do
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/r06w/plain_$i.json 2>> gpurun_out/r06w/err.txt
Y3_BENCH_FORCE_LAUNCH=1 python bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/r06w/rccl_$i.json 2>> gpurun_out/r06w/err.txt
done
