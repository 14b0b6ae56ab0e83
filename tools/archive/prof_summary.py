#!/usr/bin/env python3
"""Exports the per-kernel summary (name, calls, total/avg duration, percent) of a rocprofv3
`--kernel-trace --stats` run (rocpd sqlite output, view top_kernels) as CSV for profiles/."""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = c.execute("select * from top_kernels").fetchall()
cols = [d[0] for d in c.execute("select * from top_kernels").description]
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(cols)
    w.writerows(rows)
print("wrote", out, len(rows), "kernels")
