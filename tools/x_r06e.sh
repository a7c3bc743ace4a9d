mkdir -p gpurun_out/r06e
python -m pytest tests -x -q -m gpu > gpurun_out/r06e/gpu_tests.log 2>&1
python bench.py --steps 20 --warmup 5 --dump-ops gpurun_out/r06e/ops.txt > gpurun_out/r06e/bench.json 2> gpurun_out/r06e/bench.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06e/smoke.log 2>&1
