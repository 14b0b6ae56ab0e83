#!/bin/bash
# Round 4 evidence, run on the GPU box through gpurun: the bench lines of the four single-GPU configs (with cpu_baseline, e2e and
# parity), rocprofv3 kernel statistics of C2 and C3, PMC passes of C2 and C3 (each counter group in its own run, --kernel-trace only).
# Output under gpurun_out/r04final/, copied to profiles/r04_* by hand.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r04final"
mkdir -p "$OUT"
cd "$ROOT"
if [ "${1:-}" = tests ]; then timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6; fi
cd /tmp && export TMPDIR=/tmp
for cfg in C2 C4 C5 C3; do
  steps=50; [ $cfg = C3 ] && steps=10; [ $cfg = C5 ] && steps=20
  extra=""; [ $cfg = C2 ] && extra="--with-masking"
  timeout 900 python "$ROOT/bench.py" --config $cfg --steps $steps --warmup 10 $extra > "$OUT/bench_$cfg.json" 2> "$OUT/bench_$cfg.err"
  tail -c 300 "$OUT/bench_$cfg.err"
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c2" -o s -- python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/stats_c2.log" 2>&1
find "$OUT/stats_c2" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_C2.csv"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c3" -o s -- python "$ROOT/bench.py" --config C3 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/stats_c3.log" 2>&1
find "$OUT/stats_c3" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_C3.csv"
rm -rf "$OUT"/stats_c2 "$OUT"/stats_c3
timeout 600 "$ROOT/tools/pmc_passes.sh" C2 "$OUT/pmc_summary_C2.json" 2>&1 | tail -1
timeout 900 "$ROOT/tools/pmc_passes.sh" C3 "$OUT/pmc_summary_C3.json" 2>&1 | tail -1
cd "$ROOT" && bash tools/gpu_mask.sh > "$OUT/mask.txt" 2>&1; cp gpurun_out/mask/kernel_stats_mask.csv "$OUT/kernel_stats_mask.csv"; grep MASK_TIMING "$OUT/mask.txt"
bash tools/gpu_mask_pmc.sh > "$OUT/pmc_tantan.txt" 2>&1; cp gpurun_out/mask/pmc_tantan.json "$OUT/pmc_tantan.json"; tail -5 "$OUT/pmc_tantan.txt"
python - <<PY
import json
for c in ("C2","C4","C5","C3"):
    try:
        d=json.loads(open("$OUT/bench_%s.json"%c).read().strip().splitlines()[-1])
        e=d.get("e2e",{})
        print(c, "ms/step %.3f value %.1f parity %s | cpu hot %.3f s %.2f GCUPS | e2e %s" % (d["ms_per_step"], d["value"], d.get("parity_checked"), d["cpu_baseline"]["hot_path"]["seconds"], d["cpu_baseline"]["value"],
              {k:(round(v["reference_s"],2), round(v["ours_s"],3), round(v["speedup"],1), v["parity"]) for k,v in e.get("runs",{}).items()}))
    except Exception as ex: print(c, "failed", ex)
PY
