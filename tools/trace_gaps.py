#!/usr/bin/env python3
"""Busy time, idle gaps and per-kernel totals from a rocprofv3 --kernel-trace CSV (all streams merged).

  python tools/trace_gaps.py <dir with *kernel_trace.csv> [--skip-first-s 0]

Answers "where does a pass go beyond its dominant kernel": the union of the kernels' [start, end) intervals is the time the GPU was
doing anything at all; what is left between the first and the last kernel is launch gaps / host synchronisation.  Overlap (two
streams) shows as sum of durations > busy time."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.split("(")[0][:70]))
    rows.sort()
    if not rows:
        print("no kernel rows"); return 1
    # drop everything before the last big idle gap > 0.5 s?  no: report the whole trace and the last N windows separately
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
    gaps = []
    for s, e, _ in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    per = defaultdict(lambda: [0, 0])
    for s, e, k in rows:
        per[k][0] += e - s; per[k][1] += 1
    span = t1 - t0
    print(f"kernels {len(rows)}  span {span / 1e6:.3f} ms  busy(union) {busy / 1e6:.3f} ms  idle {(span - busy) / 1e6:.3f} ms  sum(durations) {sum(e - s for s, e, _ in rows) / 1e6:.3f} ms")
    small = [g for g in gaps if g < 200_000]
    big = [g for g in gaps if g >= 200_000]
    print(f"gaps: {len(gaps)} total; < 0.2 ms: {len(small)} summing {sum(small) / 1e6:.3f} ms (mean {sum(small) / max(1, len(small)) / 1e3:.2f} us); >= 0.2 ms: {len(big)} summing {sum(big) / 1e6:.3f} ms")
    for k, (ns, n) in sorted(per.items(), key=lambda kv: -kv[1][0])[:16]:
        print(f"  {ns / 1e6:10.3f} ms  {n:7d} x  {ns / n / 1e3:9.2f} us  {k}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
