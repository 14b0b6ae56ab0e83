#!/bin/bash
# C3 (--sensitive) check: the tests named in tools/quick_tests.txt, then kernel statistics of a short C3 bench run
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/c3"; mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest $(cat tools/quick_tests.txt) -m gpu -x -q 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python "$ROOT/bench.py" --config C3 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > "$OUT/stats.log" 2>&1
find "$OUT/stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_C3.csv"
rm -rf "$OUT/stats"
head -7 "$OUT/kernel_stats_C3.csv" | cut -c1-150
tail -1 "$OUT/stats.log" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('ms/step %.2f seed_kernel_ms %s' % (d['ms_per_step'], [round(x,2) for x in d['alone']['seed_kernel_ms']]))"
