#!/bin/bash
# last call of round 5: the whole GPU suite on the final tree (experiments compiled in, off), C3 and C2 bench lines
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r05n"; mkdir -p "$OUT"
cd "$ROOT"
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | cut -c1-600 | tail -30 > "$OUT/pytest.txt"; tail -4 "$OUT/pytest.txt"
cd /tmp && export TMPDIR=/tmp
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], 'ms/step %.3f median %.3f seed_ms %s stream in pipeline %.3f parity %s' % (d['ms_per_step'], d.get('ms_per_step_median') or 0, [round(x,3) for x in d['alone']['seed_kernel_ms']], d['roofline']['kernel_ms'], d.get('parity_checked')))" "$1"; }
timeout 600 python "$ROOT/bench.py" --config C3 --steps 10 --warmup 4 --no-e2e --no-masked-step > "$OUT/bench_C3.json" 2>/dev/null; line C3 < "$OUT/bench_C3.json"
timeout 600 python "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_C2.json" 2>/dev/null; line C2 < "$OUT/bench_C2.json"
