# Round 6 final evidence, ON THE GPU BOX (gpurun -- bash tools/r06_final_evidence.sh): rocprofv3 trace + PMC passes of the shipped binary, then three
# driver-style bench.py runs on the SAME box with the table the profile produced, then smoke(); the summaries are copied into profiles/ by hand.
mkdir -p gpurun_out/r06_final
bash tools/profile_gpu.sh r06 > gpurun_out/r06_final/prof_console.txt 2>&1
cp gpurun_out/prof_r06/traffic.json profiles/r06_traffic.json
cp gpurun_out/prof_r06/traffic.json gpurun_out/r06_final/r06_traffic.json
cp gpurun_out/prof_r06/summary.txt gpurun_out/r06_final/rocprof_summary_serial.txt
cp gpurun_out/prof_r06/kernel_stats_serial.csv gpurun_out/r06_final/kernel_stats_serial.csv
python bench.py --steps 20 --warmup 5 --dump-ops gpurun_out/r06_final/ops.txt > gpurun_out/r06_final/bench.json 2> gpurun_out/r06_final/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r06_final/bench_2.json 2>> gpurun_out/r06_final/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r06_final/bench_3.json 2>> gpurun_out/r06_final/bench.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_final/smoke.log 2>&1
