mkdir -p gpurun_out/r06b
python -m pytest tests/test_gpu_contention.py -x -q -m gpu > gpurun_out/r06b/contention.log 2>&1
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "direct_weights or private or kernel_choice or plan_options or graph" > gpurun_out/r06b/parity_subset.log 2>&1
for i in 1 2 3; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --dump-ops gpurun_out/r06b/ops_$i.txt > gpurun_out/r06b/bench_$i.json 2> gpurun_out/r06b/bench_$i.err
done
python bench.py --steps 20 --warmup 5 > gpurun_out/r06b/bench_full.json 2> gpurun_out/r06b/bench_full.err
