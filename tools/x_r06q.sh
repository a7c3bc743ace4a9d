mkdir -p gpurun_out/r06q
bash tools/profile_gpu.sh r06 > gpurun_out/r06q/prof_console.txt 2>&1
cp gpurun_out/prof_r06/traffic.json profiles/r06_traffic.json
cp gpurun_out/prof_r06/traffic.json gpurun_out/r06q/r06_traffic.json
cp gpurun_out/prof_r06/summary.txt gpurun_out/r06q/rocprof_summary_serial.txt
cp gpurun_out/prof_r06/kernel_stats_serial.csv gpurun_out/r06q/kernel_stats_serial.csv
python bench.py --steps 20 --warmup 5 --dump-ops gpurun_out/r06q/ops.txt > gpurun_out/r06q/bench.json 2> gpurun_out/r06q/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r06q/bench_2.json 2>> gpurun_out/r06q/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r06q/bench_3.json 2>> gpurun_out/r06q/bench.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06q/smoke.log 2>&1
