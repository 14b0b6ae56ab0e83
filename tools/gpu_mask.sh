#!/bin/bash
# round 4: tantan masking kernel -- parity tests, timing on 3.0e8 letters, kernel trace
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; OUT="$ROOT/gpurun_out/mask"; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_mask.py -m gpu -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python -m pytest "$ROOT/tests/test_gpu_c2_scale.py" -m gpu -x -q -s -k "mask" > "$OUT/mask.log" 2>&1
grep MASK_TIMING "$OUT/mask.log"
find "$OUT/stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_mask.csv"; rm -rf "$OUT/stats"
grep -iE "tantan|radix|scan|sort" "$OUT/kernel_stats_mask.csv" | cut -d, -f1-4 | cut -c1-150
