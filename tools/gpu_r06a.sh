#!/bin/bash
# Round 6, first GPU call (tree as round 5 left it): the counters of the configuration where Smith-Waterman dominates (C2skew: kernel
# statistics + PMC passes), and the host time line of the extension stage (DMND_TRACE=1) on C3 / C5 / C2skew -- the split of
# host_cpu_ms_per_step the device-side planning work of this round starts from.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r06a"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for cfg in C3 C2skew C5; do
  DMND_TRACE=1 timeout 600 python "$ROOT/bench.py" --config $cfg --steps 4 --warmup 2 --no-cpu-baseline --no-e2e --no-masked-step > "$OUT/trace_$cfg.json" 2> "$OUT/trace_$cfg.err"
  grep -c "dmnd_extend\[" "$OUT/trace_$cfg.err"
  grep "dmnd_extend\[" "$OUT/trace_$cfg.err" | tail -6
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_skew" -o s -- python "$ROOT/bench.py" --config C2skew --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-masked-step > "$OUT/stats_skew.log" 2>&1
find "$OUT/stats_skew" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_C2skew.csv"
rm -rf "$OUT/stats_skew"
head -12 "$OUT/kernel_stats_C2skew.csv"
PMC_SHORT=1 timeout 900 "$ROOT/tools/pmc_passes.sh" C2skew "$OUT/pmc_summary_C2skew.json" --no-e2e 2>&1 | tail -2
