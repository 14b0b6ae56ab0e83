#!/bin/bash
# Round 6: C2 with the extension on the device: extension contexts 1 / 2 / 3, and kernel statistics of device vs host extension
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r06f"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
show() {
python - "$1" "$2" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        print(sys.argv[2], {k: d.get(k) for k in ("ms_per_step", "host_cpu_ms_per_step")}, d["roofline"]["frac"], d["alone"]["batch_latency_ms"], d["alone"]["seed_stage_call_ms"], d["alone"]["extension_call_ms"], d["latency_in_pipeline"])
PY
}
for e in 1 2 3 4; do
  timeout 600 python "$ROOT/bench.py" --ext-contexts $e --no-cpu-baseline --no-e2e --no-masked-step > "$OUT/e$e.json" 2> "$OUT/e$e.err"; show "$OUT/e$e.json" "ext-contexts=$e"
done
for dev in 1 0; do
  DMND_EXTEND_DEVICE=$dev timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_dev$dev" -o s -- python "$ROOT/bench.py" --steps 50 --warmup 10 --no-cpu-baseline --no-e2e --no-masked-step > "$OUT/stats_dev$dev.log" 2>&1
  find "$OUT/stats_dev$dev" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_C2_dev$dev.csv"
  rm -rf "$OUT/stats_dev$dev"
  show "$OUT/stats_dev$dev.log" "rocprof dev=$dev"
  python - "$OUT/kernel_stats_C2_dev$dev.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print("total kernel ms", tot / 1e6, "launches", calls)
for r in rows[:14]:
    print("  %-70s calls %6s avg %9.1f us total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e6))
PY
done
