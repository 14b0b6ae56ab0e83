#!/usr/bin/env python3
"""Cost split of the seed stage's pair filter on the C2 blocks (first two --sensitive shapes): normal run, Hamming threshold
set above 48 (nothing passes: the pure all-pairs stage), and the same with the untiled kernel (DMND_SEED_TILED=0)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from diamond_amd import hip, synth, workload
db, doff, q, qoff = synth.generate(100_000, members=10, queries=10_000, seed=20260923)
qd, ql = workload.sequence_set(q, qoff); td, tl = workload.sequence_set(db, doff)
params = hip.default_params(); params.db_letters = float(doff[-1])
ctx = hip.Context(params=params)
ctx.upload_block(hip.QUERY, qd, ql); ctx.upload_block(hip.TARGET, td, tl)
sp, gf = hip.seed_params_preset("sensitive", params, threads=8)
sp.n_shapes = 2
for name, ham, tiled in (("normal", 11, None), ("hamming_off", 49, None), ("untiled", 11, "0"), ("untiled_hamming_off", 49, "0")):
    sp.hamming_filter_id = ham
    if tiled is None: os.environ.pop("DMND_SEED_TILED", None)
    else: os.environ["DMND_SEED_TILED"] = tiled
    for _ in range(2):
        hits = ctx.seed_search(sp)
    print(name, len(hits), [round(x, 1) for x in ctx.seed_kernel_ms()])
ctx.close()
