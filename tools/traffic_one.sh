#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes of one bench.py workload (run on the GPU box): tools/traffic_one.sh <tag> <bench args...>
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$REPO/gpurun_out/traffic_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extras --streams 1 --resident $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 $REPO/bench.py $ARGS > $OUT/f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 $REPO/bench.py $ARGS > $OUT/w.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for sub in ("f", "w"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(acc, key=lambda k: -acc[k].get("FETCH_SIZE", 0)):
    fe = acc[k]["FETCH_SIZE"] / max(n[k]["FETCH_SIZE"], 1); wr = acc[k]["WRITE_SIZE"] / max(n[k]["WRITE_SIZE"], 1)
    print("%-62s launches %4d  avg us %8.2f  FETCH KiB %10.1f (x2 = %8.2f MB)  WRITE KiB %10.1f  traffic MB %8.2f" % (
        k, n[k]["FETCH_SIZE"], sum(dur[k]) / max(len(dur[k]), 1), fe, fe * 2 * 1024 / 1e6, wr, (fe * 2 + wr) * 1024 / 1e6))
PY
