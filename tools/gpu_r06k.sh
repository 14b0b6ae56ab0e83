#!/bin/bash
# Round 6: kernel statistics of C2 / C3 (short) after the planner moved from scratch to LDS, and the C2 / C3 / C2skew step
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r06k"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for cfg in C2 C3 C2skew; do
  steps=50; warm=10; [ $cfg != C2 ] && steps=8 && warm=2
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/st_$cfg" -o s -- python "$ROOT/bench.py" --config $cfg --steps $steps --warmup $warm --no-cpu-baseline --no-e2e --no-masked-step > "$OUT/stats_$cfg.log" 2>&1
  find "$OUT/st_$cfg" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_$cfg.csv"
  rm -rf "$OUT/st_$cfg"
  python - "$OUT/stats_$cfg.log" "$OUT/kernel_stats_$cfg.csv" $cfg <<'PY'
import csv, json, sys
for line in open(sys.argv[1]):
    if line.startswith("{") and '"metric"' in line:
        d = json.loads(line)
        print(sys.argv[3], {k: d.get(k) for k in ("ms_per_step", "value", "host_cpu_ms_per_step")})
rows = list(csv.DictReader(open(sys.argv[2])))
for r in rows:
    if any(k in r["Name"] for k in ("plan_", "ext_", "xdrop", "hauser", "traceback")):
        print("  %-60s calls %6s avg %9.1f us total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e6))
PY
done
