#!/bin/bash
# Round 6: after the small-call trims (bias cached per query block, no host copies the device path does not need, no contended
# atomic): extension parity tests, then C2 default / host extension / two seed contexts, twice each
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r06h"
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests/test_gpu_extend.py tests/test_gpu_gapped.py tests/test_gpu_skew.py tests/test_gpu_fullscale.py tests/test_gpu_mask.py tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -5 | tee "$OUT/tests.log"
cd /tmp && export TMPDIR=/tmp
show() {
python - "$1" "$2" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        print(sys.argv[2], {k: d.get(k) for k in ("ms_per_step", "host_cpu_ms_per_step")}, round(d["roofline"]["frac"], 3), [round(d["alone"][k], 2) for k in ("batch_latency_ms", "seed_stage_call_ms", "extension_call_ms")], d["latency_in_pipeline"]["seed_stage_call_ms"]["p50"])
PY
}
for rep in 1 2; do
  timeout 600 python "$ROOT/bench.py" --no-cpu-baseline --no-e2e --no-masked-step > "$OUT/default.$rep.json" 2> "$OUT/default.$rep.err"; show "$OUT/default.$rep.json" default
  DMND_EXTEND_DEVICE=0 timeout 600 python "$ROOT/bench.py" --no-cpu-baseline --no-e2e --no-masked-step > "$OUT/exthost.$rep.json" 2> "$OUT/exthost.$rep.err"; show "$OUT/exthost.$rep.json" exthost
  timeout 600 python "$ROOT/bench.py" --seed-contexts 2 --no-cpu-baseline --no-e2e --no-masked-step > "$OUT/sc2.$rep.json" 2> "$OUT/sc2.$rep.err"; show "$OUT/sc2.$rep.json" seed-contexts=2
  timeout 600 python "$ROOT/bench.py" --seed-contexts 2 --ext-contexts 2 --no-cpu-baseline --no-e2e --no-masked-step > "$OUT/sc2e2.$rep.json" 2> "$OUT/sc2e2.$rep.err"; show "$OUT/sc2e2.$rep.json" seed-contexts=2,ext-contexts=2
done
