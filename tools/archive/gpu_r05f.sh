#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r05f"; mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests/test_gpu_skew.py -m gpu -q 2>&1 | cut -c1-1500 > "$OUT/pytest.txt"; grep -n "Error\|passed\|failed" "$OUT/pytest.txt" | head -20
cd /tmp && export TMPDIR=/tmp
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], 'ms/step %.3f median %.3f host_cpu %.1f ext %s' % (d['ms_per_step'], d.get('ms_per_step_median') or 0, d['host_cpu_ms_per_step'], d['latency_in_pipeline']['extension_of_a_batch_ms']))" "$1"; }
for et in "4 24" "4 32" "6 24" "6 36" "8 32" "4 24"; do
  set -- $et
  timeout 400 python "$ROOT/bench.py" --config C5 --steps 12 --warmup 4 --no-cpu-baseline --ext-contexts $1 --host-threads $2 2>/dev/null | line "C5 ext-contexts=$1 host-threads=$2" | tee -a "$OUT/ext_contexts.txt"
done
