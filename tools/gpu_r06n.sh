#!/bin/bash
# Round 6: the headline configuration with the sleeping-poll waits: step times, the longest step, host CPU; four runs, and DMND_SPIN_SYNC=1 twice
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r06n"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  env "$@" timeout 600 python "$ROOT/bench.py" --no-cpu-baseline --no-e2e --no-masked-step > "$OUT/$name.json" 2> "$OUT/$name.err"
  python - "$OUT/$name.json" "$name" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line); e = d["ms_each_step"]
        print(sys.argv[2], "ms/step %.3f median %.3f longest %s host_cpu %.1f frac %.3f" % (d["ms_per_step"], d["ms_per_step_median"] or 0, sorted(e)[-3:], d["host_cpu_ms_per_step"], d["roofline"]["frac"]))
PY
}
for i in 1 2 3 4; do run default$i X=1; done
run spin1 DMND_SPIN_SYNC=1; run spin2 DMND_SPIN_SYNC=1
run c3 X=1 2>/dev/null
