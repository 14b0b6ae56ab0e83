#!/bin/bash
# kernel statistics of C3 on the final tree (the command of tools/profile_r05.sh)
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r05u"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c3" -o s -- python "$ROOT/bench.py" --config C3 --steps 3 --warmup 1 --no-cpu-baseline --no-masked-step > "$OUT/stats_c3.log" 2>&1
find "$OUT/stats_c3" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_C3.csv"
rm -rf "$OUT/stats_c3"
head -12 "$OUT/kernel_stats_C3.csv" | sed 's/(.*)",/",/' | cut -d, -f1-4 | grep -v rocprim
