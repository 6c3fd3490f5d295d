#!/usr/bin/env python3
"""Summarise a profiles/run_rocprof.sh output directory: per-kernel time and PMC sums per kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats)")
for f in find("trace/**/*kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        print("  %-60s calls=%s total_ns=%s avg_ns=%s pct=%s" % (row.get("Name", "")[:60], row.get("Calls"), row.get("TotalDurationNs"),
                                                                 row.get("AverageNs"), row.get("Percentage")))
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds"):
    files = find(sub + "/**/*counter_collection.csv")
    if not files:
        print("== %s: no counter file" % sub)
        continue
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")[:48]
            acc[k][row.get("Counter_Name")] += float(row.get("Counter_Value", 0) or 0)
            cnt[(k, row.get("Counter_Name"))] += 1
    print("== %s (sum over dispatches; n = dispatches)" % sub)
    for k in sorted(acc):
        for c in sorted(acc[k]):
            print("  %-48s %-24s sum=%.6g n=%d per_dispatch=%.6g" % (k, c, acc[k][c], cnt[(k, c)], acc[k][c] / cnt[(k, c)]))
