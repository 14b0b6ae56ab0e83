#!/bin/bash
# kernel statistics of the C3 bench with the partitioned join
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/part"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python "$ROOT/bench.py" --config C3 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > "$OUT/stats.log" 2>&1
find "$OUT/stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_C3_part.csv"
rm -rf "$OUT/stats"
head -8 "$OUT/kernel_stats_C3_part.csv" | cut -c1-200
