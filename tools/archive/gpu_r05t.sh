#!/bin/bash
# classify with grouped coefficients: seed tests, C3 and C5 lines with parity inside the run
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r05t"; mkdir -p "$OUT"
cd "$ROOT"
timeout 80 python -m pytest tests/test_gpu_seed.py -m gpu -q -x 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], 'ms/step %.3f median %.3f seed_ms %s parity %s' % (d['ms_per_step'], d.get('ms_per_step_median') or 0, [round(x,3) for x in d['alone']['seed_kernel_ms']], d.get('parity_checked')))" "$1"; }
timeout 110 python "$ROOT/bench.py" --config C3 --steps 10 --warmup 4 --no-e2e --no-masked-step > "$OUT/bench_C3.json" 2>/dev/null; line C3 < "$OUT/bench_C3.json"
timeout 150 python "$ROOT/bench.py" --config C5 --steps 20 --warmup 10 --no-e2e --no-masked-step > "$OUT/bench_C5.json" 2>/dev/null; line C5 < "$OUT/bench_C5.json"
