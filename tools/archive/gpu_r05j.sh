#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r05j"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
env MODES_EXTEND=0 DMND_SEED_SJ=1 DMND_SEED_SLOTS_X8=16 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/st" -o s -- python "$ROOT/tools/seed_modes.py" sensitive 2 > "$OUT/log.txt" 2>&1
find "$OUT/st" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_sj.csv"; rm -rf "$OUT/st"
head -12 "$OUT/kernel_stats_sj.csv" | cut -d, -f1-4 | cut -c1-160
