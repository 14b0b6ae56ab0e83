#!/usr/bin/env python3
"""GPU-busy fraction of a bench.py run from a rocprofv3 --kernel-trace CSV: the union of all kernel (and copy-kernel) intervals
inside the steady part of the run divided by its length. usage: gpu_busy.py kernel_trace.csv OUT.json [skip_fraction]
The steady part = the middle of the trace (the first and last `skip_fraction` of the kernels are the warm-up / the serial
measurements that bench.py appends)."""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
lo, hi = int(len(iv) * skip), int(len(iv) * (1 - skip))
iv = iv[lo:hi]
t0, t1 = iv[0][0], max(e for _, e, _ in iv)
busy, cur_s, cur_e = 0, None, None
for s, e, _ in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
total = sum(e - s for s, e, _ in iv)
per = {}
for s, e, n in iv:
    k = n.split("(")[0][:60]
    per[k] = per.get(k, 0) + (e - s)
out = {"window_ms": (t1 - t0) / 1e6, "busy_ms": busy / 1e6, "busy_fraction": busy / (t1 - t0), "sum_of_kernel_ms": total / 1e6,
       "overlap_factor": total / busy, "kernels_in_window": len(iv),
       "top_kernels_ms": dict(sorted(((k, round(v / 1e6, 2)) for k, v in per.items()), key=lambda x: -x[1])[:10])}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
