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
gather = out["cal_gather32"]
cal = {"_source": "tools/membench.hip calib under rocprofv3 --pmc FETCH_SIZE (tools/pmc_calibrate.py); table in profiles/r02b_pmc_fetch_calibration_membench.txt",
       "_meaning": "fetch_scale / write_scale turn the counter into PHYSICAL bytes for bench.py's roofline.traffic.  Coalesced reads (16 B and 8 B per lane, "
                   "streams and long runs alike) are under-reported by exactly 2 (known / counted = 2.00): scale 2.  Random 32-byte records are counted at 64 B each "
                   "(known / counted = 0.50): the counter already charges whole sectors, i.e. at least what the memory system moved, so it is used as is (scale 1).",
       "patterns": out,
       "walk": {"fetch_scale": 1.0, "write_scale": 1.0, "pattern": "cal_gather32",
                "counted_bytes_per_32B_record": round(gather["fetch_size_bytes"] / (gather["known_bytes"] / 32), 2),
                "gather_ceiling_Ggathers_per_s": round(gather["GBps"] / 32, 2),
                "_note": "random 32-byte gathers from a 16 GiB table saturate at this rate on this chip (useful bytes: %.0f GB/s); the walk is bound by it, not by streaming bandwidth" % gather["GBps"]},
       "expand": {"fetch_scale": round(out["cal_runs8/1024"]["scale"], 3), "write_scale": 1.0, "pattern": "cal_runs8/1024",
                  "_write": "WRITE_SIZE is calibrated by expand_kernel itself: 12.000 B per 12-byte tuple (profiles/r01_pmc_config3_full_summary.txt)"},
       "retain": {"fetch_scale": 1.0, "write_scale": 1.0, "pattern": "cal_gather32 (16-byte probes + 4-byte child lists: sector-granular like the walk)"}}
if "--write" in sys.argv:
    i = sys.argv.index("--write")
    p = sys.argv[i + 1] if i + 1 < len(sys.argv) else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_calibration.json")
    json.dump(cal, open(p, "w"), indent=1)
    print("wrote", p)
