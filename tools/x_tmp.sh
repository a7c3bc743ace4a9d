mkdir -p gpurun_out/r06z4
for i in 1 2 3; do
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r06z4/bench_new_$i.json 2>> gpurun_out/r06z4/err.txt
Y3_HIP_LIB=$PWD/pytorch-yolov3_amd/lib/libyolov3_hip_oldwait.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r06z4/bench_old_$i.json 2>> gpurun_out/r06z4/err.txt
done
