#!/usr/bin/env python3
"""FETCH_SIZE calibration for the product kernels' access patterns (MI355X_MICROARCH.md, HBM section:
"calibrate on a known byte count in your own access pattern before trusting an absolute").

  cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -o pmc -- $REPO/tools/membench calib > DIR/calib.txt
  python tools/pmc_calibrate.py DIR [--write]

Each `cal_*` kernel of tools/membench.hip reads a known number of bytes once; the ratio known / counted is the
scale factor bench.py applies to FETCH_SIZE of the kernel with that pattern:
  walk    <- cal_gather32   (one random 32-byte record per lane)
  expand  <- cal_runs8/1024 (8 bytes per lane, consecutive lanes on consecutive entries of long runs)
--write stores them in profiles/pmc_calibration.json.
"""
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
known = []
for line in open(os.path.join(d, "calib.txt")):
    if line.startswith("CAL "):
        _, name, b = line.split()
        known.append((name, int(b)))
rows = []
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if row.get("Counter_Name") == "FETCH_SIZE" and "cal_" in row["Kernel_Name"]:
            rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"], float(row["Counter_Value"]) * 1024,
                         int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
rows.sort()
assert len(rows) == len(known), (len(rows), len(known))
out = {}
print("%-26s %16s %16s %8s %10s" % ("pattern", "known bytes", "FETCH_SIZE bytes", "scale", "GB/s"))
for (name, b), (_, kn, counted, ns) in zip(known, rows):
    scale = b / counted if counted else float("nan")
    out[name] = {"known_bytes": b, "fetch_size_bytes": counted, "scale": round(scale, 4), "GBps": round(b / ns, 1)}
    print("%-26s %16d %16.0f %8.3f %10.1f" % (name, b, counted, scale, b / ns))
cal = {"_source": "tools/membench.hip calib under rocprofv3 --pmc FETCH_SIZE (tools/pmc_calibrate.py)", "patterns": out,
       "walk": {"fetch_scale": out["cal_gather32"]["scale"], "write_scale": 1.0, "pattern": "cal_gather32"},
       "expand": {"fetch_scale": out["cal_runs8/1024"]["scale"], "write_scale": 1.0, "pattern": "cal_runs8/1024",
                  "_write": "WRITE_SIZE is calibrated by expand_kernel itself: 12.000 B per 12-byte tuple (profiles/r01_pmc_config3_full_summary.txt)"},
       "retain": {"fetch_scale": out["cal_gather32"]["scale"], "write_scale": 1.0, "pattern": "cal_gather32 (16-byte probes + 4-byte child lists: upper bound)"}}
if "--write" in sys.argv:
    i = sys.argv.index("--write")
    p = sys.argv[i + 1] if i + 1 < len(sys.argv) else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_calibration.json")
    json.dump(cal, open(p, "w"), indent=1)
    print("wrote", p)
