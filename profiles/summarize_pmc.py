#!/usr/bin/env python3
"""Per-kernel FETCH_SIZE / WRITE_SIZE summary from collect_pmc.sh's two rocprofv3 passes."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
res = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(out, counter, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
            res[k][counter][0] += 1
            res[k][counter][1] += float(row["Counter_Value"])
print("%-42s %8s %18s %18s   (values are the counters' native unit: KiB; per dispatch = total/dispatches)" % ("kernel", "disp", "FETCH_SIZE total", "WRITE_SIZE total"))
for k, v in sorted(res.items(), key=lambda kv: -(kv[1]["FETCH_SIZE"][1] + kv[1]["WRITE_SIZE"][1])):
    print("%-42s %8d %18.1f %18.1f" % (k, max(v["FETCH_SIZE"][0], v["WRITE_SIZE"][0]), v["FETCH_SIZE"][1], v["WRITE_SIZE"][1]))
