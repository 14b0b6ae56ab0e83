#!/bin/bash
# round 4: bench test with the masked step + C2 bench line with --with-masking
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; OUT="$ROOT/gpurun_out/r4a"; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_bench.py -m gpu -x -q -k "e2e_and_hot" 2>&1 | tail -5
timeout 1200 python bench.py --with-masking > "$OUT/bench_C2_masked.json" 2> "$OUT/bench_C2_masked.err"; tail -3 "$OUT/bench_C2_masked.err"
python - "$OUT/bench_C2_masked.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("ms_per_step", d["ms_per_step"], "value", d["value"], "parity", d.get("parity_checked"))
print("masked_step", json.dumps(d.get("masked_step"))[:1500])
print("e2e", {k: (round(v["speedup"], 2), v["ours_runs_s"], v["parity"]) for k, v in d["e2e"]["runs"].items()})
PY
