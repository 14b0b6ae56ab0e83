#!/bin/bash
# round 4 (scratch): three mask_block calls on the 3.0e8-letter block (no planted repeats): first-call cost vs steady state
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
cat > /tmp/exp.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["ROOT"]); sys.path.insert(0, os.path.join(os.environ["ROOT"], "tests"))
import numpy as np, torch
from diamond_amd import hip, synth, workload
db, doff, q, qoff = synth.generate(100_000, members=10, queries=10_000, seed=20260923)
td, tl = workload.sequence_set(db, doff)
assert hip.load().dmnd_init(0) == 0
ctx = hip.Context()
for rep in range(3):
    ctx.upload_block(hip.TARGET, td, tl)
    masked = td.copy()
    n = ctx.mask_block(hip.TARGET, masked)
    print("call", rep, "masked", int(n), "kernel_ms", ctx.mask_kernel_ms(), flush=True)
ctx.close()
PY
ROOT="$ROOT" timeout 300 python /tmp/exp.py 2>&1 | tail -3
