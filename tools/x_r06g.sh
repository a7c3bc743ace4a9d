# profile the shipped binary (kernel trace + PMC passes), then bench.py on the SAME box with the table it produced.
# The raw rocprofv3 csv files exceed gpurun's 64 MiB copy-back limit: only the summaries leave the box.
bash tools/profile_gpu.sh r06 > gpurun_out/prof_r06_console.txt 2>&1
cp gpurun_out/prof_r06/traffic.json profiles/r06_traffic.json
mkdir -p gpurun_out/r06g
cp gpurun_out/prof_r06/traffic.json gpurun_out/r06g/r06_traffic.json
cp gpurun_out/prof_r06/summary.txt gpurun_out/r06g/rocprof_summary_serial.txt
find gpurun_out/prof_r06/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r06g/kernel_stats_serial.csv
for f in trace pmc_fetch pmc_write pmc_sq; do tail -3 gpurun_out/prof_r06/$f.log > gpurun_out/r06g/$f.log.tail 2>/dev/null; done
rm -rf gpurun_out/prof_r06
for i in 1 2 3; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r06g/bench_$i.json 2> gpurun_out/r06g/bench_$i.err
done
