#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + stats of bench.py, then separate
# PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-extras $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $ARGS > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py $ARGS > $OUT/pmc_sq.log 2>&1
find $OUT -name "*.csv" | head -50
python3 $REPO/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
tail -60 $OUT/summary.txt
python3 $REPO/tools/make_traffic_table.py $OUT $OUT/traffic.json
