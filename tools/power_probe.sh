# Samples GPU power / clocks (rocm-smi) while bench.py runs: tools/power_probe.sh [bench args...]
python bench.py --no-extras --no-cpu-baseline --steps 6000 --warmup 20 "$@" > gpurun_out/power_bench.log 2>&1 &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|GPU use" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 0.5
done | grep -v "use (%): 0" | awk 'NR % 4 == 1' | head -12
tail -1 gpurun_out/power_bench.log | cut -c1-200
