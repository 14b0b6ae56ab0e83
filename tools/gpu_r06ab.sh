#!/bin/bash
# kernel statistics of one config, the seed stage's smaller kernels: tools/gpu_r06ab.sh C3|C2skew
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r06ab"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cfg=${1:-C3}
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$cfg" -o s -- python "$ROOT/bench.py" --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-masked-step --seed-contexts 1 > "$OUT/stats_$cfg.log" 2>&1
find "$OUT/stats_$cfg" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_$cfg.csv"
rm -rf "$OUT/stats_$cfg"
grep "seed_leftmost\|seed_score\|seed_deferred\|seed_collect\|seed_pair" "$OUT/kernel_stats_$cfg.csv" | cut -c1-60,100-260
