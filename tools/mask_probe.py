#!/usr/bin/env python3
"""Where the masking of one C2 database block spends its time (round 5): tantan (dmnd_mask_block) and motif soft masking
(dmnd_soft_mask_block) timed apart on the 3.0e8-letter block, host wall clock per call; with DMND_TRACE=1 the laps of
dmnd_mask_block go to stderr; under `rocprofv3 --kernel-trace --stats` the kernels' own times. usage: tools/mask_probe.py [runs]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diamond_amd import hip, synth, workload

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
db, doff, q, qoff = synth.generate(100_000, members=10, queries=10_000, seed=20260923)
td, tl = workload.sequence_set(db, doff)
qd, ql = workload.sequence_set(q, qoff)
hip.load_motif_table()
params = hip.default_params()
raw, mc = hip.Context(params=params), hip.Context(params=params)
for c in (raw, mc):
    c.upload_block(hip.QUERY, qd, ql)
    c.upload_block(hip.TARGET, td, tl)
td_m = td.copy()
for r in range(runs):
    mc.copy_block(hip.TARGET, raw)
    t0 = time.perf_counter()
    n = mc.mask_block(hip.TARGET, td_m)
    t1 = time.perf_counter()
    mc.soft_mask_block(hip.TARGET)
    t2 = time.perf_counter()
    print("MASK_PROBE run %d: tantan call %.2f ms (kernel %.2f), motif soft masking call %.2f ms, masked letters %d" % (r, (t1 - t0) * 1e3, mc.mask_kernel_ms(), (t2 - t1) * 1e3, n), flush=True)
raw.close(); mc.close()
