#!/bin/bash
# load_query without the copy + sort of hits that arrive in order: C3 line with parity inside the run, host CPU per step
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r05w"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 62 python "$ROOT/bench.py" --config C3 --steps 10 --warmup 4 --no-e2e --no-masked-step > "$OUT/bench_C3.json" 2>/dev/null
python -c "
import json
d=json.loads(open('$OUT/bench_C3.json').read().strip().splitlines()[-1]); print('C3 ms/step %.3f host cpu %.1f parity %s ext alone %.1f' % (d['ms_per_step'], d['host_cpu_ms_per_step'], d.get('parity_checked'), d['alone']['extension_call_ms']))"
