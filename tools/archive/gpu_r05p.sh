#!/bin/bash
# round 5: the deferred pass's gather through a folded need map in LDS -- size of the folded map
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r05p"; mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python -m pytest tests/test_gpu_seed.py -m gpu -q -x -k "folded or tiled" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], 'ms/step %.3f | seed call alone %.3f | seed_ms %s' % (d['ms_per_step'], d['alone']['seed_stage_call_ms'], [round(x,3) for x in d['alone']['seed_kernel_ms']]))" "$1"; }
for v in 13 14 off; do
  if [ $v = off ]; then export DMND_SEED_COLLECT_FOLDED_FROM=1000000000000; else export DMND_SEED_NEED_FOLD_LOG2=$v; fi
  DMND_TRACE=1 timeout 400 python "$ROOT/bench.py" --config C3 --steps 3 --warmup 2 --no-e2e --no-masked-step --no-cpu-baseline --no-pipeline > "$OUT/bench_C3_fold$v.json" 2> "$OUT/trace_$v.txt"; line "C3 fold 2^$v words" < "$OUT/bench_C3_fold$v.json"
done
grep -m1 "deferred:" "$OUT/trace_13.txt" | cut -c1-900
