#!/usr/bin/env python3
"""Seed stage of one sensitivity preset on the C2 blocks (10k queries x 1M sequences): per-kernel device time and the
number of joined reference positions per shape (DMND_TRACE=1 prints them). usage: tools/seed_modes.py default [runs]"""
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
for _ in range(runs):
    hits = ctx.seed_search(sp)
    print(mode, len(hits), ctx.seed_kernel_ms())
ctx.close()
