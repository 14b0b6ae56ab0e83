#!/bin/bash
# skew test again; extension-context sweep for C5 and C3 (how many batches are extended at the same time, host threads split among them)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r05e"; mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests/test_gpu_skew.py -m gpu -q 2>&1 | cut -c1-1500 > "$OUT/pytest.txt"; grep -n "Error\|assert\|passed\|failed" "$OUT/pytest.txt" | head -20
cd /tmp && export TMPDIR=/tmp
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], 'ms/step %.3f median %.3f host_cpu %.1f ext %s' % (d['ms_per_step'], d.get('ms_per_step_median') or 0, d['host_cpu_ms_per_step'], d['latency_in_pipeline']['extension_of_a_batch_ms']))" "$1"; }
for e in 2 3 4; do
  timeout 400 python "$ROOT/bench.py" --config C5 --steps 9 --warmup 3 --no-cpu-baseline --ext-contexts $e 2>/dev/null | line "C5 ext-contexts=$e" | tee -a "$OUT/ext_contexts.txt"
done
timeout 400 python "$ROOT/bench.py" --config C5 --steps 9 --warmup 3 --no-cpu-baseline --ext-contexts 4 --host-threads 24 2>/dev/null | line "C5 ext-contexts=4 host-threads=24" | tee -a "$OUT/ext_contexts.txt"
for e in 2 3 4; do
  timeout 400 python "$ROOT/bench.py" --config C3 --steps 8 --warmup 2 --no-cpu-baseline --no-masked-step --ext-contexts $e 2>/dev/null | line "C3 ext-contexts=$e" | tee -a "$OUT/ext_contexts.txt"
done
