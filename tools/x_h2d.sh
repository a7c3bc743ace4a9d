#!/bin/bash
# PCIe-inclusive variant under rocprofv3: kernel + memory-copy timeline (no PMC)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/h2d_trace
rm -rf $OUT; mkdir -p $OUT
python3 $REPO/tools/h2d_probe.py 2>&1 | grep -v amdgpu
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- python3 $REPO/bench.py --h2d --steps 12 --warmup 6 --no-cpu-baseline --no-extras > $OUT/bench.log 2>&1
tail -c 400 $OUT/bench.log
find $OUT -name "*.csv" | head
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*memory_copy_trace.csv", recursive=True)
k = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    print(len(rows), "copies; columns", list(rows[0].keys()))
    big = [r for r in rows if int(r.get("Bytes", r.get("Size", 0)) or 0) > 1000000] if ("Bytes" in rows[0] or "Size" in rows[0]) else rows
    for r in rows[-40:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(r.get("Direction"), r.get("Bytes", r.get("Size")), "dur us %.1f" % ((e - s) / 1e3), "start %.3f ms" % (s / 1e6 % 100000))
if k:
    rows = list(csv.DictReader(open(k[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[len(rows)//2]["Start_Timestamp"]); t1 = int(rows[-1]["End_Timestamp"])
    print("second half of the kernel trace: %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(rows) - len(rows)//2))
PY
