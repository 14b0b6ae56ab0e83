"""Host half of the extension stage on a CPU-only box: dmnd_extend_plan (Hauser bias, load_hits, x-drop ungapped extension, greedy
chaining, band merging) on the seed hits of a bench configuration (tools/probes/dump_seed_hits.py), timed, with a digest of the
plan so that a change of the host code can be A/B-compared.
  python tools/probes/plan_bench.py gpurun_out/r05v C3 [threads] [repeats]"""
import hashlib
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                      # noqa: E402
from diamond_amd import hip       # noqa: E402

src, cfg = sys.argv[1], sys.argv[2]
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 8
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
c = bench.CONFIGS[cfg]
w = bench.Workload(cfg, c.get("families", 100_000), c.get("queries", 10_000), 1, 0, "db")
params = hip.default_params()
params.db_letters = float(w.db_letters)
hits = np.load(os.path.join(src, "hits_%s.npy" % cfg))
td, tl = w.blocks[0][2], w.blocks[0][3]
best = 1e9
for _ in range(reps):
    t0 = time.perf_counter(); c0 = time.process_time()
    cbs, plan = hip.extend_plan(params, w.qd, w.ql, td, tl, hits, threads=threads, query_contexts=w.contexts)
    dt, cpu = time.perf_counter() - t0, time.process_time() - c0
    best = min(best, dt)
    print("%s: %d hits -> %d DpTargets, %.1f ms wall, %.1f ms CPU (%d threads)" % (cfg, len(hits), len(plan), dt * 1e3, cpu * 1e3, threads))
order = np.lexsort((plan["d_end"], plan["d_begin"], plan["target"], plan["query"]))
print("plan digest", hashlib.md5(plan[order].tobytes()).hexdigest(), "bias digest", hashlib.md5(cbs.tobytes()).hexdigest())
