mkdir -p gpurun_out/r06v
for i in 1 2; do
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/r06v/plain_$i.json 2>> gpurun_out/r06v/err.txt
Y3_BENCH_FORCE_LAUNCH=1 python bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/r06v/rccl_$i.json 2>> gpurun_out/r06v/err.txt
done
