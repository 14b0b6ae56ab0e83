#!/usr/bin/env python3
"""Experiment: C2 hot path with E extension contexts in flight (batches s, s+1, ... extended concurrently), seed stage on its own
context. usage: tools/pipe_probe.py E STEPS HOST_THREADS   (env DMND_EXTEND_SPLIT / DMND_EXTEND_RUNNERS select the runner layout)"""
import concurrent.futures as cf
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diamond_amd import hip, synth, workload  # noqa: E402

E, K, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
db, doff, q, qoff = synth.generate(100_000, members=10, queries=10_000, seed=20260923)
qd, ql = workload.sequence_set(q, qoff)
td, tl = workload.sequence_set(db, doff)
params = hip.default_params()
params.db_letters = float(doff[-1])


def make():
    c = hip.Context(device=0, params=params)
    c.upload_block(hip.QUERY, qd, ql)
    c.upload_block(hip.TARGET, td, tl)
    return c


ctx_seed = make()
ctx_ext = [make() for _ in range(E)]
sp = hip.seed_params_fast(threads=8)
seed_pool = cf.ThreadPoolExecutor(1)
ext_pools = [cf.ThreadPoolExecutor(1) for _ in range(E)]      # one thread per extension context: a context runs one call at a time


def seed():
    torch.cuda.set_device(0)
    return ctx_seed.seed_search(sp)


def ext(k, fut):
    torch.cuda.set_device(0)
    hits = fut.result()
    m, _ = ctx_ext[k % E].extend(qd, td, hits, threads=T)
    return m.size


def run(n, primed):
    seeds = [primed] + [seed_pool.submit(seed) for _ in range(n)]      # n seed stages inside this call (the last one only awaited)
    exts = [ext_pools[k % E].submit(ext, k, seeds[k]) for k in range(n)]
    r = [e.result() for e in exts]
    return seeds[n], r


primed = seed_pool.submit(seed)
primed, _ = run(3 * E, primed)
primed.result()
torch.cuda.synchronize()
for c in ctx_ext + [ctx_seed]:
    c.touch_streams()
t0, c0 = time.perf_counter(), time.process_time()
primed, r = run(K, primed)
primed.result()
torch.cuda.synchronize()
dt, cpu = time.perf_counter() - t0, time.process_time() - c0
print("E=%d threads=%d split=%s runners=%s swipe32=%s: %.3f ms/step, cpu %.1f ms/step, matches %d" % (
    E, T, os.environ.get("DMND_EXTEND_SPLIT"), os.environ.get("DMND_EXTEND_RUNNERS"), os.environ.get("DMND_SWIPE32"), dt / K * 1e3, cpu / K * 1e3, r[-1]))
