#!/bin/bash
# short check on the GPU box: the tests named in tools/quick_tests.txt (one pytest argument per line), then the C2 bench line with e2e
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; mkdir -p gpurun_out/quick
timeout 900 python -m pytest $(cat tools/quick_tests.txt) -m gpu -x -q 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
timeout 600 python "$ROOT/bench.py" --steps 20 --warmup 3 > "$ROOT/gpurun_out/quick/bench_C2.json" 2> "$ROOT/gpurun_out/quick/bench_C2.err"
tail -c 400 "$ROOT/gpurun_out/quick/bench_C2.err"
python - <<PY
import json
d=json.loads(open("$ROOT/gpurun_out/quick/bench_C2.json").read().strip().splitlines()[-1])
e=d.get("e2e",{})
print("C2 ms/step %.3f value %.1f parity %s | e2e %s" % (d["ms_per_step"], d["value"], d.get("parity_checked"),
      {k:(round(v["reference_s"],2), round(v["ours_s"],3), round(v["speedup"],1), v["parity"]) for k,v in e.get("runs",{}).items()}))
PY
