#!/bin/bash
# bench lines of C2 and C5 again (defaults: 50 steps after 10 warm-up steps), into gpurun_out/r03final
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r03final"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 python "$ROOT/bench.py" > "$OUT/bench_C2.json" 2> "$OUT/bench_C2.err"
timeout 600 python "$ROOT/bench.py" --config C5 --steps 20 > "$OUT/bench_C5.json" 2> "$OUT/bench_C5.err"
timeout 600 python "$ROOT/bench.py" --config C4 > "$OUT/bench_C4.json" 2> "$OUT/bench_C4.err"
python - <<PY
import json
for c in ("C2","C5","C4"):
    d=json.loads(open("$OUT/bench_%s.json"%c).read().strip().splitlines()[-1])
    e=d.get("e2e",{})
    print(c, "steps", d["steps"], "ms/step %.3f (median of windows %.3f) value %.1f parity %s | e2e %s" % (d["ms_per_step"], d["ms_per_step_median_of_3_step_windows"], d["value"], d.get("parity_checked"),
          {k:(round(v["reference_s"],2), round(v["ours_s"],3), round(v["speedup"],1), v["parity"]) for k,v in e.get("runs",{}).items()}))
    print("   each", [round(x,1) for x in d["ms_each_step"]][:50])
PY
