mkdir -p gpurun_out/r06y
for i in 1 2; do
Y3_BENCH_FORCE_LAUNCH=1 python bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/r06y/rccl_$i.json 2>> gpurun_out/r06y/err.txt
Y3_X_NO_GATHER=1 Y3_BENCH_FORCE_LAUNCH=1 python bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/r06y/nogather_$i.json 2>> gpurun_out/r06y/err.txt
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/r06y/plain_$i.json 2>> gpurun_out/r06y/err.txt
done
