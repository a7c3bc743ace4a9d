#!/bin/bash
# rocprofv3 PMC passes over tools/block_bench.py (fused 1x1 -> 3x3 kernel; run on the GPU box): $1 = shape filter, rest = extra args
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_block
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/p1 -- python3 $REPO/tools/block_bench.py --only "$1" --rounds 1 --iters 5 ${@:2} > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/p2 -- python3 $REPO/tools/block_bench.py --only "$1" --rounds 1 --iters 5 ${@:2} > $OUT/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAVES --output-format csv -d $OUT/p3 -- python3 $REPO/tools/block_bench.py --only "$1" --rounds 1 --iters 5 ${@:2} > $OUT/p3.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in ("p1", "p2", "p3"):
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
