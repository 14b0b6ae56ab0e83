#!/bin/bash
# Round 6: the device half of the extension stage (extend_kernels.hip) -- the extension parity tests and the full-size runs, then
# host CPU per step and the step time of C3 / C2skew / C5 / C2 with it on and off (DMND_EXTEND_DEVICE=0) on one box
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/${R06D_OUT:-r06d}"
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests/test_gpu_extend.py tests/test_gpu_gapped.py tests/test_gpu_skew.py tests/test_gpu_fullscale.py tests/test_gpu_xdrop.py tests/test_gpu_edge_cases.py -m gpu -x -q 2>&1 | tail -15 | tee "$OUT/tests.log"
cd /tmp && export TMPDIR=/tmp
for cfg in C3 C2skew C5 C2; do
  for dev in 1 0; do
    DMND_EXTEND_DEVICE=$dev DMND_TRACE=1 timeout 600 python "$ROOT/bench.py" --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-masked-step > "$OUT/trace_${cfg}_dev$dev.json" 2> "$OUT/trace_${cfg}_dev$dev.err"
    echo "== $cfg device=$dev"
    grep -A1 "dmnd_extend\[" "$OUT/trace_${cfg}_dev$dev.err" | tail -2
    grep "dmnd_extend total" "$OUT/trace_${cfg}_dev$dev.err" | tail -2
    python - "$OUT/trace_${cfg}_dev$dev.json" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        print({k: d.get(k) for k in ("ms_per_step", "value", "host_cpu_ms_per_step", "host_cpu_ms_per_step_by_thread", "parity_checked")})
PY
  done
done
