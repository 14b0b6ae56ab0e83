#!/bin/bash
# kernel statistics of C2skew (sweep kernels: calls, total, average) on the current tree
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r06u"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for cfg in C2skew; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$cfg" -o s -- python "$ROOT/bench.py" --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-masked-step > "$OUT/stats_$cfg.log" 2>&1
  find "$OUT/stats_$cfg" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_$cfg.csv"
  rm -rf "$OUT/stats_$cfg"
  grep "swipe16\|traceback" "$OUT/kernel_stats_$cfg.csv" | cut -c1-70,200-420 | sed 's/signed char.*int)//'
done
cd "$ROOT"
for rep in 1 2; do timeout 900 python bench.py --config C2skew --steps 40 --warmup 5 --no-e2e --no-masked-step --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('C2skew ms/step %.3f' % d['ms_per_step'])
"; done
