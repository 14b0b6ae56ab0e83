"""Seed hits of a bench configuration, saved for host-side work on a box without a GPU (tools/probes/plan_bench.py): the blocks are
regenerated there from the same seeds (bench.Workload), the hits come from dmnd_seed_search here.
  python tools/probes/dump_seed_hits.py OUTDIR C2 C3"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                      # noqa: E402
from diamond_amd import hip       # noqa: E402

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
for cfg in sys.argv[2:]:
    c = bench.CONFIGS[cfg]
    w = bench.Workload(cfg, c.get("families", 100_000), c.get("queries", 10_000), 1, 0, "db")
    params = hip.default_params()
    params.db_letters = float(w.db_letters)
    sp, gf = w.seed_params(params)
    ctx = hip.Context(device=0, params=params)
    ctx.upload_block(hip.QUERY, w.qd, w.ql)
    ctx.upload_block(hip.TARGET, w.blocks[0][2], w.blocks[0][3])
    ctx.set_query_contexts(w.contexts)
    hits = ctx.seed_search(sp)
    np.save(os.path.join(out, "hits_%s.npy" % cfg), hits)
    print(cfg, len(hits), "hits", hits.nbytes >> 20, "MiB")
    ctx.close()
