#!/bin/bash
# Round 5, first GPU call: the whole GPU suite on the rebuilt tree (one clear launch per search, pooled tantan scratch, range join,
# cbs passes), the C2 bench line with the masked step and alternating blocks, C3 with and without non-temporal stream loads,
# and the PMC passes of C5 (short set). Output under gpurun_out/r05a/.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r05a"; mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > "$OUT/pytest.txt"; tail -3 "$OUT/pytest.txt"
cd /tmp && export TMPDIR=/tmp
timeout 600 python "$ROOT/bench.py" --steps 30 --warmup 6 > "$OUT/bench_C2.json" 2> "$OUT/bench_C2.err"; tail -c 300 "$OUT/bench_C2.err"
timeout 300 python "$ROOT/bench.py" --steps 30 --warmup 6 --same-block --no-cpu-baseline --no-masked-step > "$OUT/bench_C2_same_block.json" 2> "$OUT/bench_C2_same_block.err"
for nt in 0 1; do
  DMND_SEED_STREAM_NT=$nt timeout 600 python "$ROOT/bench.py" --config C3 --steps 6 --warmup 2 --no-cpu-baseline > "$OUT/bench_C3_nt$nt.json" 2> "$OUT/bench_C3_nt$nt.err"; tail -c 200 "$OUT/bench_C3_nt$nt.err"
done
PMC_SHORT=1 timeout 900 "$ROOT/tools/pmc_passes.sh" C5 "$OUT/pmc_summary_C5.json" 2>&1 | tail -2
python - <<PY
import json
for f in ("bench_C2","bench_C2_same_block","bench_C3_nt0","bench_C3_nt1"):
    try:
        d=json.loads(open("$OUT/%s.json"%f).read().strip().splitlines()[-1])
        print(f, "ms/step %.3f median %s value %.1f parity %s seed_ms %s" % (d["ms_per_step"], d.get("ms_per_step_median"), d["value"], d.get("parity_checked"), {k:round(v,3) for k,v in d["seed_kernel_ms"].items()}))
        if "masked_step" in d: print("  masked", d["masked_step"]["ms_per_step"], d["masked_step"]["parts_ms"], d["masked_step"].get("parity"))
        if "e2e" in d: print("  e2e", {k:(round(v["speedup"],2), round(v["speedup_min"],2), v["parity"]) for k,v in d["e2e"]["runs"].items()})
        print("  host_cpu_ms_per_step", d.get("host_cpu_ms_per_step"), "kernel_ms", d["roofline"]["kernel_ms"], d["roofline"]["kernel_ms_alone"])
    except Exception as ex: print(f, "failed", ex)
PY
