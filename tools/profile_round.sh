#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace statistics and separate PMC passes of bench.py, written under
# gpurun_out/<tag>/ ; tools/pmc_summary.py + the kernel_stats.csv are then copied into profiles/ by hand.
# usage: tools/profile_round.sh TAG   (PMC passes never combine with other trace domains)
set -u
TAG="${1:-r01}"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o "$TAG" -- $BENCH > "$OUT/stats.log" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  name=$(echo "$grp" | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc_$name" -o pmc -- python "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_$name.log" 2>&1
done
python "$ROOT/tools/pmc_summary.py" "$OUT/pmc_summary.json" "$OUT"/pmc_*/ > "$OUT/pmc_summary.log" 2>&1
find "$OUT" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
ls -la "$OUT"
