#!/bin/bash
# Round 2 evidence, run on the GPU box through gpurun: bench lines of the three single-GPU configs, rocprofv3 kernel statistics of
# C2 and C3, and PMC passes of C2 (each its own run with --kernel-trace only). Output under gpurun_out/r02/, copied to profiles/ by hand.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r02"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for cfg in C2 C4 C3; do
  steps=20; [ $cfg = C3 ] && steps=5
  timeout 900 python "$ROOT/bench.py" --config $cfg --steps $steps --warmup 2 > "$OUT/bench_$cfg.json" 2> "$OUT/bench_$cfg.err"
  tail -c 400 "$OUT/bench_$cfg.err"
done
B2="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c2" -o s -- $B2 > "$OUT/stats_c2.log" 2>&1
find "$OUT/stats_c2" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_C2.csv"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c3" -o s -- python "$ROOT/bench.py" --config C3 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/stats_c3.log" 2>&1
find "$OUT/stats_c3" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_C3.csv"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc$i" -o pmc -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline > "$OUT/pmc$i.log" 2>&1
done
python "$ROOT/tools/pmc_summary.py" "$OUT/pmc_summary_C2.json" "$OUT"/pmc*/ > "$OUT/pmc_summary.log" 2>&1
rm -rf "$OUT"/stats_c2 "$OUT"/stats_c3 "$OUT"/pmc?/
ls -la "$OUT"
head -c 1500 "$OUT/bench_C2.json"
