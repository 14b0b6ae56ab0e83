#!/usr/bin/env python3
"""Seed stage of one sensitivity preset on the C2 blocks (10k queries x 1M sequences): per-kernel device time and the
number of joined reference positions per shape (DMND_TRACE=1 prints them), then the extension stage on those hits. usage: tools/seed_modes.py default [runs]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DMND_TRACE", "1")
from diamond_amd import hip, synth, workload

mode = sys.argv[1] if len(sys.argv) > 1 else "default"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
db, doff, q, qoff = synth.generate(100_000, members=10, queries=10_000, seed=20260923)
qd, ql = workload.sequence_set(q, qoff)
td, tl = workload.sequence_set(db, doff)
params = hip.default_params()
params.db_letters = float(doff[-1])
ctx = hip.Context(params=params)
ctx.upload_block(hip.QUERY, qd, ql)
ctx.upload_block(hip.TARGET, td, tl)
sp, gf = hip.seed_params_preset(mode, params, threads=8)
import time
ctx.set_gapped_filter(gf)
for _ in range(runs):
    t0 = time.perf_counter()
    hits = ctx.seed_search(sp)
    t1 = time.perf_counter()
    print(mode, len(hits), "seed_ms %.2f" % ((t1 - t0) * 1e3), ctx.seed_kernel_ms(), flush=True)
    if os.environ.get("MODES_EXTEND", "1") != "0":
        t1 = time.perf_counter()
        m, _ = ctx.extend(qd, td, hits, threads=int(os.environ.get("MODES_THREADS", "32")))
        t2 = time.perf_counter()
        print(mode, "extend_ms %.2f" % ((t2 - t1) * 1e3), len(m), {k: round(v, 2) for k, v in ctx.extend_stats().items()}, flush=True)
ctx.close()
