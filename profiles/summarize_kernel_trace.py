#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, share) of a `rocprofv3 --kernel-trace --output-format csv` run.
usage: summarize_kernel_trace.py <output dir>"""
import csv
import glob
import os
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0, 0, 1 << 62, 0])
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        d = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        a = acc[k]
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in acc.values()) or 1
print("%-60s %8s %16s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
for k, a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-60s %8d %16d %12.0f %12d %12d %7.2f" % (k[:60], a[0], a[1], a[1] / a[0], a[2], a[3], 100.0 * a[1] / tot))
