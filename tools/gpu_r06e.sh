#!/bin/bash
# Round 6: the headline configuration (C2, default flags of bench.py) under the wait / extension variants of this round, same box
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r06e"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  for rep in 1 2; do
    env "$@" timeout 600 python "$ROOT/bench.py" --no-cpu-baseline --no-e2e --no-masked-step > "$OUT/$name.$rep.json" 2> "$OUT/$name.$rep.err"
    python - "$OUT/$name.$rep.json" "$name" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        print(sys.argv[2], {k: d.get(k) for k in ("ms_per_step", "steps", "host_cpu_ms_per_step", "ms_per_step_median")}, d["roofline"]["frac"], d["alone"]["batch_latency_ms"], d["alone"]["seed_stage_call_ms"], d["alone"]["extension_call_ms"])
PY
  done
}
run default X=1
run spin DMND_SPIN_SYNC=1
run exthost DMND_EXTEND_DEVICE=0
run exthost_spin DMND_EXTEND_DEVICE=0 DMND_SPIN_SYNC=1
run poll50 DMND_SYNC_SPIN_US=50
run poll500 DMND_SYNC_SPIN_US=500
