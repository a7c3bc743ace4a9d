#!/bin/bash
# rocprofv3 PMC pass over tools/conv_bench.py for one layer (run on the GPU box)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_conv
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u | tr '\n' ' ' > $OUT/sq_counters.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/p1 -- python3 $REPO/tools/conv_bench.py --only "$1" --rounds 1 --iters 5 ${@:2} > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/p2 -- python3 $REPO/tools/conv_bench.py --only "$1" --rounds 1 --iters 5 ${@:2} > $OUT/p2.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in ("p1", "p2"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not fs:
        print("no counters for", sub); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    for k in acc:
        if "conv" in k:
            print(k, " ".join("%s=%.4g" % (c, acc[k][c] / n[k][c]) for c in sorted(acc[k])))
PY
