#!/bin/bash
# C5 line of the final tree with the step count of profiles/r05_bench_C5.json (20 timed steps: five clumps of four batches)
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r05s"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 python "$ROOT/bench.py" --config C5 --steps 20 --warmup 10 --no-e2e --no-masked-step --no-cpu-baseline > "$OUT/bench_C5.json" 2>/dev/null
python -c "
import sys,json
d=json.loads(open('$OUT/bench_C5.json').read().strip().splitlines()[-1]); print('C5 ms/step %.3f median %.3f seed_ms %s host cpu %.1f' % (d['ms_per_step'], d.get('ms_per_step_median') or 0, [round(x,3) for x in d['alone']['seed_kernel_ms']], d['host_cpu_ms_per_step']))"
