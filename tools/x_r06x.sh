mkdir -p gpurun_out/r06x
cd /tmp && export TMPDIR=/tmp
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 Y3_BENCH_FORCE_DIST=1
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06x/trace -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline --no-extras --sustain 0 --profile-passes 3 > $GRAFT_REPO_ROOT/gpurun_out/r06x/bench.log 2>&1
cd $GRAFT_REPO_ROOT/gpurun_out/r06x
find trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} kernel_stats.csv
find trace -name "*memory_copy_stats.csv" | head -1 | xargs -I{} cp {} memory_copy_stats.csv
find trace -name "*memory_copy_trace.csv" | head -1 | xargs -I{} sh -c 'head -40 {} > memory_copy_trace_head.csv'
rm -rf trace
