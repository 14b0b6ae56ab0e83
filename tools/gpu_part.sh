#!/bin/bash
# partitioned join of the short-seed pipeline: parity tests, then kernel statistics of the C3 bench with it
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/part"; mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python -m pytest tests/test_gpu_seed.py -m gpu -x -q 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
DMND_TRACE=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python "$ROOT/bench.py" --config C3 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > "$OUT/stats.log" 2>&1
grep -m3 "partitioned join" "$OUT/stats.log"
find "$OUT/stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_C3_part.csv"
rm -rf "$OUT/stats"
head -4 "$OUT/kernel_stats_C3_part.csv" | cut -c1-160
tail -1 "$OUT/stats.log" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('ms/step %.2f seed_kernel_ms %s' % (d['ms_per_step'], [round(x,2) for x in d['alone']['seed_kernel_ms']]))"
