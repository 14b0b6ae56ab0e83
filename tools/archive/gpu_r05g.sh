#!/bin/bash
# the two-lane short-seed pipeline (stage 2 of shape s beside index + stream of shape s + 1): parity, then timing with and without
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r05g"; mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests/test_gpu_seed.py tests/test_gpu_fullscale.py tests/test_gpu_cli.py tests/test_gpu_extend.py -m gpu -q -k "seed or c3 or sensitive or extend or query_indexed" 2>&1 | cut -c1-1200 | tail -30 > "$OUT/pytest.txt"; tail -5 "$OUT/pytest.txt"
cd /tmp && export TMPDIR=/tmp
for ov in 0 1 1; do
  env MODES_EXTEND=0 DMND_SEED_OVERLAP=$ov timeout 200 python "$ROOT/tools/seed_modes.py" sensitive 3 2>/dev/null | grep -v "^dmnd" | tail -2 | sed "s/^/overlap=$ov : /" | tee -a "$OUT/overlap.txt"
done
for m in very-sensitive mid-sensitive; do for ov in 0 1; do
  env MODES_EXTEND=0 DMND_SEED_OVERLAP=$ov timeout 300 python "$ROOT/tools/seed_modes.py" $m 2 2>/dev/null | grep -v "^dmnd" | tail -1 | sed "s/^/overlap=$ov : /" | tee -a "$OUT/overlap.txt"
done; done
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], 'ms/step %.3f median %.3f seed_ms %s parity %s' % (d['ms_per_step'], d.get('ms_per_step_median') or 0, [round(x,2) for x in d['alone']['seed_kernel_ms']], d.get('parity_checked')))" "$1"; }
timeout 600 python "$ROOT/bench.py" --config C3 --steps 8 --warmup 2 --no-e2e --no-masked-step > "$OUT/bench_C3.json" 2>/dev/null; line "C3 overlap" < "$OUT/bench_C3.json"
DMND_SEED_OVERLAP=0 timeout 600 python "$ROOT/bench.py" --config C3 --steps 8 --warmup 2 --no-cpu-baseline --no-masked-step 2>/dev/null | line "C3 no overlap"
