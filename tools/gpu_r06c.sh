#!/bin/bash
# Round 6: where the host CPU time of a step goes -- per OS thread (bench.py host_cpu_ms_per_step_by_thread) and per phase of
# dmnd_extend (DMND_TRACE=1: process CPU time of the serial steps after the timed region)
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r06c"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for cfg in C3 C2skew C5 C2; do
  DMND_TRACE=1 timeout 600 python "$ROOT/bench.py" --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-masked-step > "$OUT/trace_$cfg.json" 2> "$OUT/trace_$cfg.err"
  echo "== $cfg"
  grep -A1 "dmnd_extend\[" "$OUT/trace_$cfg.err" | tail -5
  python - "$OUT/trace_$cfg.json" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        print({k: d.get(k) for k in ("ms_per_step", "host_cpu_ms_per_step", "host_cpu_ms_per_step_by_thread")})
PY
done
