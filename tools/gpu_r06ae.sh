#!/bin/bash
# refresh of ONE config's committed evidence (kernel statistics, short PMC set, bench line): tools/gpu_r06ae.sh C2skew
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r06final"; mkdir -p "$OUT"
cfg=${1:-C2skew}
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_x" -o s -- python "$ROOT/bench.py" --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-masked-step > "$OUT/stats_x.log" 2>&1
find "$OUT/stats_x" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_$cfg.csv"
rm -rf "$OUT/stats_x"
PMC_SHORT=1 timeout 900 "$ROOT/tools/pmc_passes.sh" $cfg "$OUT/pmc_summary_$cfg.json" --no-e2e 2>&1 | tail -1
cp "$OUT/kernel_stats_$cfg.csv" "$ROOT/profiles/r06_kernel_stats_$cfg.csv"; cp "$OUT/pmc_summary_$cfg.json" "$ROOT/profiles/r06_pmc_summary_$cfg.json"
cd "$ROOT"
extra=""; [ $cfg = C2skew ] && extra="--no-e2e"
timeout 900 python bench.py --config $cfg --steps 30 --warmup 10 $extra > "$OUT/bench_$cfg.json" 2> "$OUT/bench_$cfg.err"
python - <<PY
import json
d=json.loads(open("$OUT/bench_$cfg.json").read().strip().splitlines()[-1])
print("$cfg", d["summary"])
PY
