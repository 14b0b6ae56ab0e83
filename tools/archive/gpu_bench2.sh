#!/bin/bash
# bench lines again into gpurun_out/r03final (defaults: 50 steps after 10 warm-up steps): tools/bench2_cases.txt names the configs
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r03final"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in $(cat "$ROOT/tools/bench2_cases.txt"); do
  steps=50; [ $c = C3 ] && steps=10; [ $c = C5 ] && steps=20
  timeout 600 python "$ROOT/bench.py" --config $c --steps $steps > "$OUT/bench_$c.json" 2> "$OUT/bench_$c.err"
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$c.json").read().strip().splitlines()[-1])
e=d.get("e2e",{})
print("$c", "steps", d["steps"], "ms/step %.3f (median of windows %.3f) value %.1f parity %s finish %.2f | e2e %s" % (d["ms_per_step"], d["ms_per_step_median_of_3_step_windows"], d["value"], d.get("parity_checked"), d["alone"]["finish_ms"],
      {k:(round(v["reference_s"],2), round(v["ours_s"],3), round(v["speedup"],1), v["parity"]) for k,v in e.get("runs",{}).items()}))
PY
done
