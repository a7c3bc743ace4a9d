#!/usr/bin/env python3
"""Concurrency picture of a rocprofv3 --kernel-trace of bench.py (several batches in flight): how much of the wall time
has 0 / 1 / 2 / 3+ kernels running, and per kernel family: launches, summed duration, duration when it ran ALONE vs
overlapped.  Usage: python3 tools/timeline.py <dir with *kernel_trace.csv> [skip_fraction]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def fam(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.search(r"\d+([a-z0-9_]+_kernel)", name)
    base = m.group(1) if m else name.split("(")[0].split("<")[0]
    if "halo_ws" in name:
        m = re.search(r"Li(\d+)ELi(\d+)E", name)
        if m:
            base += "_ns%s_mi%s" % (m.group(1), m.group(2))
    return base[:44]


def main():
    d = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fam(r["Kernel_Name"])) for r in csv.DictReader(open(f))]
    rows.sort()
    t0 = rows[0][0] + (rows[-1][1] - rows[0][0]) * skip        # skip warm-up / set-up
    rows = [r for r in rows if r[0] >= t0]
    ev = []
    for s, e, k in rows:
        ev.append((s, 1, k))
        ev.append((e, -1, k))
    ev.sort()
    conc = defaultdict(float)
    alone = defaultdict(float)
    total = defaultdict(float)
    n = defaultdict(int)
    active = defaultdict(int)
    cur, last = 0, ev[0][0]
    for t, dlt, k in ev:
        dt = (t - last) / 1e3
        conc[min(cur, 4)] += dt
        live = [a for a in active if active[a] > 0]
        for a in live:
            total[a] += dt * active[a]
            if cur == 1:
                alone[a] += dt
        if dlt > 0:
            n[k] += 1
        active[k] += dlt
        cur += dlt
        last = t
    wall = sum(conc.values())
    print("wall %.1f us over %d kernels; time with k kernels running: %s" % (
        wall, len(rows), "  ".join("%d: %.1f%%" % (k, 100 * v / wall) for k, v in sorted(conc.items()))))
    print("%-46s %7s %10s %10s %8s" % ("kernel family", "calls", "busy us", "alone us", "alone %"))
    for k in sorted(total, key=lambda k: -total[k]):
        print("%-46s %7d %10.1f %10.1f %7.1f%%" % (k, n[k], total[k], alone[k], 100 * alone[k] / max(total[k], 1e-9)))


if __name__ == "__main__":
    main()
