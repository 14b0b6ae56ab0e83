#!/bin/bash
# Round 6 evidence, run on the GPU box through gpurun: the whole GPU suite, the bench lines of the configs (cpu_baseline, e2e, parity,
# masked step), rocprofv3 kernel statistics of C2 and C3, PMC passes of C2, C3 (full sets) and C4, C5 (short sets; tools/pmc_passes.sh).
# Output under gpurun_out/r06final/, copied to profiles/r06_* by hand.   tools/profile_r06.sh [tests] [nopmc]
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r06final"
mkdir -p "$OUT"
cd "$ROOT"
if [[ " $* " == *" tests "* ]]; then timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -30 > "$OUT/pytest.txt"; tail -4 "$OUT/pytest.txt"; fi
cd /tmp && export TMPDIR=/tmp
if [[ " $* " == *" bench "* ]]; then
for cfg in C2 C4 C5 C3 C2skew; do
  steps=50; [ $cfg = C3 ] && steps=30; [ $cfg = C5 ] && steps=32; [ $cfg = C2skew ] && steps=30
  extra=""; [ $cfg = C2skew ] && extra="--no-e2e"
  timeout 900 python "$ROOT/bench.py" --config $cfg --steps $steps --warmup 10 $extra > "$OUT/bench_$cfg.json" 2> "$OUT/bench_$cfg.err"
  tail -c 300 "$OUT/bench_$cfg.err"
done
fi
if [[ " $* " == *" counters "* ]]; then
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c2" -o s -- python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-masked-step > "$OUT/stats_c2.log" 2>&1
find "$OUT/stats_c2" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_C2.csv"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c3" -o s -- python "$ROOT/bench.py" --config C3 --steps 3 --warmup 1 --no-cpu-baseline --no-masked-step > "$OUT/stats_c3.log" 2>&1
find "$OUT/stats_c3" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_C3.csv"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c5" -o s -- python "$ROOT/bench.py" --config C5 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/stats_c5.log" 2>&1
find "$OUT/stats_c5" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_C5.csv"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_skew" -o s -- python "$ROOT/bench.py" --config C2skew --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-masked-step > "$OUT/stats_skew.log" 2>&1
find "$OUT/stats_skew" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_C2skew.csv"
rm -rf "$OUT"/stats_c2 "$OUT"/stats_c3 "$OUT"/stats_c5 "$OUT"/stats_skew
fi
if [[ " $* " == *" counters "* && " $* " != *" nopmc "* ]]; then
  timeout 600 "$ROOT/tools/pmc_passes.sh" C2 "$OUT/pmc_summary_C2.json" 2>&1 | tail -1
  timeout 900 "$ROOT/tools/pmc_passes.sh" C3 "$OUT/pmc_summary_C3.json" 2>&1 | tail -1
  PMC_SHORT=1 timeout 600 "$ROOT/tools/pmc_passes.sh" C5 "$OUT/pmc_summary_C5.json" 2>&1 | tail -1
  PMC_SHORT=1 timeout 400 "$ROOT/tools/pmc_passes.sh" C4 "$OUT/pmc_summary_C4.json" 2>&1 | tail -1
  PMC_SHORT=1 timeout 900 "$ROOT/tools/pmc_passes.sh" C2skew "$OUT/pmc_summary_C2skew.json" --no-e2e 2>&1 | tail -1
fi
python - <<PY
import json
for c in ("C2","C4","C5","C3","C2skew"):
    try:
        d=json.loads(open("$OUT/bench_%s.json"%c).read().strip().splitlines()[-1])
        e=d.get("e2e",{})
        print(c, "ms/step %.3f median %s value %.1f parity %s | cpu hot %.3f s %.2f GCUPS | host_cpu %.1f | e2e %s" % (d["ms_per_step"], d.get("ms_per_step_median"), d["value"], d.get("parity_checked"), d["cpu_baseline"]["hot_path"]["seconds"], d["cpu_baseline"]["value"], d["host_cpu_ms_per_step"],
              {k:(round(v["reference_s"],2), round(v["ours_s"],3), round(v["speedup"],1), round(v["speedup_min"],1), v["parity"]) for k,v in e.get("runs",{}).items()}))
        if "masked_step" in d: print("   masked", round(d["masked_step"]["ms_per_step"],2), round(d["masked_step"]["stages_back_to_back"]["ms_per_step"],2), d["masked_step"]["stages_back_to_back"]["parts_ms"], d["masked_step"].get("records_equal_back_to_back"), d["masked_step"].get("parity",{}).get("matches"))
        if "scaling_model" in d: print("   scaling", {k:(v["decomposition"], round(v["predicted_speedup"],2)) for k,v in d["scaling_model"]["best"].items()})
        print("   roofline", {k: d["roofline"].get(k) for k in ("frac", "traffic", "hbm_measured_frac", "l2_requests_frac")}, "| sweep", d.get("sweep_roofline", {}).get("frac"))
    except Exception as ex: print(c, "failed", ex)
PY
