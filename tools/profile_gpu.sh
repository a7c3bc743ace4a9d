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
# serial steps (one stream) with the plan options of the default three-stream bench run (Y3_AM_DEFAULT | Y3_AM_HALO_TILE256 =
# 0xb429d = 737949), so that every kernel runs the layers it runs in the headline measurement
TUNE="--streams 1 --tuning auto_mask=737949"
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-extras $TUNE $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $ARGS > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py $ARGS > $OUT/pmc_sq.log 2>&1
# the other single-GPU workloads of BASELINE.json: traffic passes only (bench.py other_configs[*].roofline.traffic)
for W in "yolov3-tiny 416 8 float32" "yolov3-spp 608 16 bf16" "yolov3 608 16 float32" "yolov3 608 16 fp16"; do
  set -- $W
  KEY=$1_$2_b$3_$4
  WARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extras $TUNE --model $1 --dim $2 --batch $3 --dtype $4"
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$KEY -- python3 $REPO/bench.py $WARGS > $OUT/pmc_fetch_$KEY.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$KEY -- python3 $REPO/bench.py $WARGS > $OUT/pmc_write_$KEY.log 2>&1
done
find $OUT -name "*.csv" | head -50
python3 $REPO/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
tail -60 $OUT/summary.txt
python3 $REPO/tools/make_traffic_table.py $OUT $OUT/traffic.json
# the raw rocprofv3 csv files (a trace + seven counter passes of ~1000 steps) exceed gpurun's 64 MiB copy-back limit: keep the
# kernel-stats table, the summary and the traffic table, drop the rest (KEEP_RAW=1 keeps everything: run it where the size does not matter)
if [ "${KEEP_RAW:-0}" != "1" ]; then
  find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_serial.csv
  for d in trace pmc_fetch pmc_write pmc_sq; do tail -3 $OUT/$d.log > $OUT/$d.log.tail 2>/dev/null; done
  find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
  rm -f $OUT/*.log
fi

