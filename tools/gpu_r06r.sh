#!/bin/bash
# kernel statistics of the sweeps with the row classes on (C2skew, C2) + the instruction counters of the row kernels
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r06r"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i "lds\|SQ_INST_CYCLES\|SQ_ACTIVE_INST\|SQ_WAIT" | head -60 > "$OUT/avail.txt"
for cfg in C2skew C2; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$cfg" -o s -- python "$ROOT/bench.py" --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-masked-step > "$OUT/stats_$cfg.log" 2>&1
  find "$OUT/stats_$cfg" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_$cfg.csv"
  rm -rf "$OUT/stats_$cfg"
  grep "swipe16\|traceback" "$OUT/kernel_stats_$cfg.csv" | cut -c1-70,200-420 | sed 's/signed char.*int)//'
done
PMC_SHORT=1 timeout 900 "$ROOT/tools/pmc_passes.sh" C2skew "$OUT/pmc_summary_C2skew.json" --no-e2e 2>&1 | tail -1
