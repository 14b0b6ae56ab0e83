#!/bin/bash
# Round 6, second GPU call: the device planner of the extension stage (plan_kernels.hip) -- the extension parity tests, then the host
# time line (DMND_TRACE=1) and host_cpu_ms_per_step of C3 / C2skew / C5 with the planner on and off (DMND_EXTEND_PLAN_GPU=0) on one box.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r06b"
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests/test_gpu_extend.py tests/test_gpu_gapped.py tests/test_gpu_skew.py tests/test_gpu_fullscale.py tests/test_gpu_xdrop.py -m gpu -x -q 2>&1 | tail -15 | tee "$OUT/tests.log"
cd /tmp && export TMPDIR=/tmp
for cfg in C3 C2skew C5; do
  for plan in 1 0; do
    DMND_EXTEND_PLAN_GPU=$plan DMND_TRACE=1 timeout 600 python "$ROOT/bench.py" --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-masked-step > "$OUT/trace_${cfg}_plan$plan.json" 2> "$OUT/trace_${cfg}_plan$plan.err"
    echo "== $cfg plan=$plan"
    grep "dmnd_extend\[" "$OUT/trace_${cfg}_plan$plan.err" | tail -3
    python - "$OUT/trace_${cfg}_plan$plan.json" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        print({k: d.get(k) for k in ("ms_per_step", "host_cpu_ms_per_step", "parity_checked", "value")}, d.get("extend_plan"))
PY
  done
done
