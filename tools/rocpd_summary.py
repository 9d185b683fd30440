#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats) into a per-kernel CSV: calls, total/avg/min/max ns, %.
usage: python tools/rocpd_summary.py gpurun_out/prof/x_results.db profiles/out.csv"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
    for r in rows:
        w.writerow([r[0], r[1], int(r[2]), round(r[3], 1), int(r[4]), int(r[5]), round(100.0 * r[2] / tot, 3)])
for r in rows[:14]:
    print(f"{r[0][:60]:60s} calls={r[1]:6d} total_ms={r[2]/1e6:9.3f} avg_us={r[3]/1e3:9.2f} {100.0*r[2]/tot:5.1f}%")
