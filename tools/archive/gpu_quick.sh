#!/bin/bash
# short check on the GPU box: the tests named in tools/quick_tests.txt (arguments of one pytest command line), then kernel statistics
# and the bench line of C2
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; OUT="$ROOT/gpurun_out/quick"; mkdir -p "$OUT"
eval "timeout 900 python -m pytest $(cat tools/quick_tests.txt) -m gpu -x -q" 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > "$OUT/stats.log" 2>&1
find "$OUT/stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_C2.csv"; rm -rf "$OUT/stats"
grep -E "xdrop|stream_fast|swipe16" "$OUT/kernel_stats_C2.csv" | cut -d, -f1-4 | cut -c1-120
timeout 600 python "$ROOT/bench.py" --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('C2 steps %d ms/step %.3f median-of-windows %.3f value %.1f alone %s' % (d['steps'], d['ms_per_step'], d['ms_per_step_median_of_3_step_windows'], d['value'], {k:round(v,2) for k,v in d['alone'].items() if isinstance(v,float)}))"
