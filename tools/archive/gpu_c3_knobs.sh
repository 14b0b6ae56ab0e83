#!/bin/bash
# C3 seed stage under a few knobs of the fused path (no partitioned join): table size, non-temporal letter loads
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/part"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  env DMND_SEED_PART=0 "$@" timeout 300 python "$ROOT/bench.py" --config C3 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$*', 'ms/step %.2f seed_kernel_ms %s' % (d['ms_per_step'], [round(x,2) for x in d['alone']['seed_kernel_ms']]))"
}
run DMND_SEED_SLOTS_X8=32
run DMND_SEED_SLOTS_X8=64
