#!/usr/bin/env python3
"""Aggregates rocprofv3 `--pmc` CSV output (one pass per counter group, *_counter_collection.csv) into per-kernel,
per-launch averages: tools/pmc_summary.py OUT.json DIR [DIR ...]. FETCH_SIZE / WRITE_SIZE are reported in bytes
(the counters tick in KiB... rocprofv3 reports them in units of 1 KB, see MI355X_MICROARCH.md) and FETCH_SIZE is
also given x2 (gfx950 under-reports wide coalesced reads by 2x: upper bound)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, dirs = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, set()]))
for d in dirs:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"]
                if not k.startswith(("dmnd::", "void dmnd::")):
                    continue
                a = acc[k][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1].add(row["Dispatch_Id"])
res = {}
for k, counters in acc.items():
    e = {}
    for name, (total, launches) in counters.items():
        n = max(1, len(launches))
        if name in ("FETCH_SIZE", "WRITE_SIZE"):
            e[name + "_bytes_per_launch"] = total * 1024.0 / n
            if name == "FETCH_SIZE":
                e["FETCH_SIZE_x2_bytes_per_launch"] = 2.0 * total * 1024.0 / n
        else:
            e[name + "_per_launch"] = total / n
        e["launches"] = n
    res[k] = e
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print("wrote", out, len(res), "kernels")
