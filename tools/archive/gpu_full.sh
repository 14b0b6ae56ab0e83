#!/bin/bash
# the whole -m gpu suite as the driver runs it at round end, then the C2 bench line
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; OUT="$ROOT/gpurun_out/full"; mkdir -p "$OUT"
timeout 2400 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -12 | tee "$OUT/gpu_tests.txt"
timeout 600 python "$ROOT/bench.py" > "$OUT/bench_C2.json" 2> "$OUT/bench_C2.err"
tail -1 "$OUT/bench_C2.json" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('C2 steps %d ms/step %.3f value %.1f parity %s roofline %s' % (d['steps'], d['ms_per_step'], d['value'], d.get('parity_checked'), d['roofline'].get('frac'))); print({k:(v.get('speedup') if isinstance(v,dict) else v) for k,v in d.get('e2e',{}).get('runs',{}).items()})"
