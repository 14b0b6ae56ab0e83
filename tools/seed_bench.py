#!/usr/bin/env python3
"""Times the HIP seed stage on the C2-shaped blocks (10k queries x 1M sequences, --fast configuration)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from diamond_amd import hip, synth, workload

fam = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
db, doff, q, qoff = synth.generate(fam, members=10, queries=nq, seed=20260923)
qd, ql = workload.sequence_set(q, qoff)
td, tl = workload.sequence_set(db, doff)
ctx = hip.Context()
ctx.upload_block(hip.QUERY, qd, ql)
ctx.upload_block(hip.TARGET, td, tl)
p = hip.seed_params_fast(threads=8)
for it in range(4):
    t = time.perf_counter()
    hits = ctx.seed_search(p)
    dt = time.perf_counter() - t
    print(json.dumps({"iter": it, "wall_ms": dt * 1e3, "kernel_ms": ctx.seed_kernel_ms(), "hits": int(hits.size),
                      "ref_letters": int(doff[-1]), "query_letters": int(qoff[-1]),
                      "stream_GBps_letters_only": doff[-1] / (ctx.seed_kernel_ms()[1] * 1e-3) / 1e9}))
