#!/bin/bash
# Round 5, last call: the whole GPU suite on the final tree, then the bench lines whose defaults changed since tools/profile_r05.sh ran
# (C5: four extension contexts; C2: one hit-sort pass) and the driver's own default command line.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r05h"; mkdir -p "$OUT"
cd "$ROOT"
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | cut -c1-600 | tail -30 > "$OUT/pytest.txt"; tail -4 "$OUT/pytest.txt"
cd /tmp && export TMPDIR=/tmp
timeout 900 python "$ROOT/bench.py" --config C5 --steps 20 --warmup 10 > "$OUT/bench_C5.json" 2> "$OUT/bench_C5.err"; tail -c 200 "$OUT/bench_C5.err"
timeout 900 python "$ROOT/bench.py" --steps 50 --warmup 10 > "$OUT/bench_C2.json" 2> "$OUT/bench_C2.err"; tail -c 200 "$OUT/bench_C2.err"
timeout 600 python "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_C2_driver_line.json" 2>/dev/null
python - <<PY
import json
for c in ("C5","C2","C2_driver_line"):
    try:
        d=json.loads(open("$OUT/bench_%s.json"%c).read().strip().splitlines()[-1])
        e=d.get("e2e",{})
        print(c, "ms/step %.3f median %s value %.1f parity %s | host_cpu %.1f | roofline frac %.3f traffic %s | e2e %s" % (d["ms_per_step"], d.get("ms_per_step_median"), d["value"], d.get("parity_checked"), d["host_cpu_ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic_source"),
              {k:(round(v["speedup"],1), round(v["speedup_min"],1), v["parity"]) for k,v in e.get("runs",{}).items()}))
        if "masked_step" in d: print("   masked", round(d["masked_step"]["ms_per_step"],2), d["masked_step"].get("parity",{}).get("matches"))
        if "scaling_model" in d: print("   scaling", {k:round(v,2) for k,v in d["scaling_model"]["predicted_speedup"].items()}, d["scaling_model"].get("predicted_speedup_with_8_cpus_per_rank"))
    except Exception as ex: print(c, "failed", ex)
PY
