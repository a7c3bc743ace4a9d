"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into one small text table."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pattern):
    hits = glob.glob(os.path.join(out, sub, "**", pattern), recursive=True)
    return hits[0] if hits else None


def short(name):
    for tok in ("void ", "(anonymous namespace)::"):
        name = name.replace(tok, "")
    name = name.split("(")[0]
    return name[:90]


stats = find("trace", "*kernel_stats.csv")
if stats:
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    with open(stats) as fh:
        rows = list(csv.DictReader(fh))
    print("%-92s %8s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for r in rows[:40]:
        print("%-92s %8s %12.1f %12.2f %7s" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                               float(r["AverageNs"]) / 1e3, r["Percentage"]))

trace = find("trace", "*kernel_trace.csv")
dur = defaultdict(list)
if trace:
    with open(trace) as fh:
        for r in csv.DictReader(fh):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)

for sub, counters in (("pmc_fetch", ["FETCH_SIZE"]), ("pmc_write", ["WRITE_SIZE"]),
                      ("pmc_sq", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
                                  "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"])):
    f = find(sub, "*counter_collection.csv")
    if not f:
        print("no counter file for", sub)
        continue
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(lambda: defaultdict(int))
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k][r["Counter_Name"]] += 1
    print("\n== %s: per-launch mean of %s ==" % (sub, ", ".join(counters)))
    for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
        vals = ["%s=%.4g" % (c, acc[k][c] / max(n[k][c], 1)) for c in counters if c in acc[k]]
        print("%-92s launches=%d %s" % (k, max(n[k].values()), " ".join(vals)))
