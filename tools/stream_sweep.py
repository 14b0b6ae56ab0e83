#!/usr/bin/env python3
"""Level-1 filter sweep of the reference stream kernel on the C2 blocks (GPU box): size of the L2-resident blocked Bloom filter,
2 or 3 bits per seed, non-temporal letter loads. Prints per variant the stream kernel's time alone and the hit count (must not change).
usage: tools/stream_sweep.py [C2|C3]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
from diamond_amd import hip  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
w = bench.Workload(cfg, 100_000, 10_000, 1, 0, "db")
params = hip.default_params()
params.db_letters = float(w.db_letters)
sp, gf = w.seed_params(params)
ctx = hip.Context(params=params)
ctx.upload_block(hip.QUERY, w.qd, w.ql)
ctx.upload_block(hip.TARGET, w.td, w.tl)
ctx.set_query_contexts(w.contexts)
base = None
variants = [dict(), dict(DMND_SEED_LEVEL2="0")] + [dict(DMND_SEED_PROBE_POLICY=str(pol)) for pol in (1, 2, 3, 16, 17, 18)] \
    + [dict(DMND_SEED_BM1_KB=str(kb), DMND_SEED_BM1_K=str(k), DMND_SEED_STREAM_NT=str(nt), DMND_SEED_PROBE_POLICY=str(pol))
       for kb in (2048, 3072, 4096, 8192) for k in (2, 3) for pol in (0, 2, 16) for nt in (0, 1)]
for v in variants:
    for k in ("DMND_SEED_BM1_KB", "DMND_SEED_BM1_K", "DMND_SEED_STREAM_NT", "DMND_SEED_PROBE_POLICY", "DMND_SEED_LEVEL2"):
        os.environ.pop(k, None)
    os.environ.update(v)
    best, hits = None, None
    for _ in range(4):
        t0 = time.perf_counter()
        h = ctx.seed_search(sp)
        wall = (time.perf_counter() - t0) * 1e3
        ms = ctx.seed_kernel_ms()
        if best is None or ms[1] < best[1]:
            best = list(ms) + [wall]
        hits = h
    key = np.sort(hits, order=["query", "subject", "seed_offset"]) if hits.dtype.names else np.sort(hits)
    if base is None:
        base = key
    same = key.shape == base.shape and (key == base).all()
    print("%-110s stream %.3f ms  index %.3f  total %.3f  call %.2f  hits %d  %s" % (v or "default", best[1], best[0], best[4], best[5], hits.size, "same" if same else "DIFFERENT"), flush=True)
ctx.close()
# band geometry of the round-1 DpTargets of this workload (what the sweep kernels get): widths and lane use per band class
os.environ.pop("DMND_SEED_BM1_KB", None)
_, plan = hip.extend_plan(params, w.qd, w.ql, w.td, w.tl, hits, threads=8, query_contexts=w.contexts)
band = (plan["d_end"] - plan["d_begin"]).astype(np.int64)
print("DpTargets", plan.size, "band width percentiles (5,25,50,75,95,99):", np.percentile(band, [5, 25, 50, 75, 95, 99]).tolist())
for lo, hi in ((0, 32), (32, 64), (64, 96), (96, 128), (128, 192), (192, 256), (256, 384), (384, 512), (512, 1 << 30)):
    m = (band > lo) & (band <= hi)
    print("  band (%d, %d]: %6d targets, %5.1f %%" % (lo, hi, int(m.sum()), 100.0 * m.mean()))
