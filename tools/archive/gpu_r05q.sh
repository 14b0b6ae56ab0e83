#!/bin/bash
# round 5: stage-2 window scores from registers (seed_score_kernel), the deferred pass's gather through a folded need map
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r05q"; mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_seed.py tests/test_gpu_extend.py tests/test_gpu_skew.py -m gpu -q -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o c3 -- python "$ROOT/bench.py" --config C3 --steps 4 --warmup 2 --no-e2e --no-masked-step --no-cpu-baseline > "$OUT/bench_C3_prof.json" 2>/dev/null
f=$(ls "$OUT"/prof/*/*kernel_stats.csv "$OUT"/prof/*kernel_stats.csv 2>/dev/null | head -1); head -14 "$f" | sed 's/(.*)",/",/' | cut -d, -f1-4 ; cp "$f" "$OUT/kernel_stats_C3.csv"; rm -rf "$OUT/prof"
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], 'ms/step %.3f median %.3f | seed call alone %.3f in pipeline p50 %.3f | seed_ms %s' % (d['ms_per_step'], d.get('ms_per_step_median') or 0, d['alone']['seed_stage_call_ms'], d['latency_in_pipeline']['seed_stage_call_ms']['p50'], [round(x,3) for x in d['alone']['seed_kernel_ms']]))" "$1"; }
timeout 400 python "$ROOT/bench.py" --config C3 --steps 10 --warmup 4 --no-e2e --no-masked-step --no-cpu-baseline > "$OUT/bench_C3.json" 2>/dev/null; line "C3" < "$OUT/bench_C3.json"
timeout 400 python "$ROOT/bench.py" --config C2 --steps 60 --warmup 10 --no-e2e --no-masked-step --no-cpu-baseline > "$OUT/bench_C2.json" 2>/dev/null; line "C2" < "$OUT/bench_C2.json"
