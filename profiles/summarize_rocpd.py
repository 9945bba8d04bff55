#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into the per-kernel stats
table committed under profiles/.  usage: summarize_rocpd.py <results.db> [> out.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows) or 1
print("%-72s %8s %16s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
for r in rows:
    print("%-72s %8d %16d %12.0f %12d %12d %7.2f" % (r[0][:72], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
