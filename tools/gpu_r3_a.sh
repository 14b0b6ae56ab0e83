#!/bin/bash
# round 3, first GPU call: the changed host paths (loader, upload, exchange, launcher), the bench line with e2e, C3 counters, C5
set -u
ROOT="$GRAFT_REPO_ROOT"; OUT="$ROOT/gpurun_out/r03"; mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_bench.py tests/test_gpu_db_shard.py tests/test_gpu_cli.py tests/test_gpu_seed.py -m gpu -x -q 2>&1 | tail -15
timeout 600 python bench.py --config C2 --steps 20 --warmup 5 > "$OUT/bench_C2.json" 2> "$OUT/bench_C2.err"; tail -c 600 "$OUT/bench_C2.err"
timeout 600 python bench.py --config C5 --steps 5 --warmup 2 --no-e2e > "$OUT/bench_C5.json" 2> "$OUT/bench_C5.err"; tail -c 600 "$OUT/bench_C5.err"
timeout 900 tools/pmc_passes.sh C3 "$OUT/pmc_summary_C3.json" 2>&1 | tail -3
timeout 600 tools/pmc_passes.sh C2 "$OUT/pmc_summary_C2_before.json" 2>&1 | tail -3
python - <<PY
import json
for c in ("C2","C5"):
    try:
        d=json.loads(open("$OUT/bench_%s.json"%c).read().strip().splitlines()[-1])
        print(c, "ms/step", d["ms_per_step"], "value", d["value"], "parity", d.get("parity_checked"), "upload_ms", d["block_upload_ms"])
        print("  seed_kernel_ms", d["seed_kernel_ms"]); print("  cpu", {k:d["cpu_baseline"][k] for k in ("value","hot_path","whole_process")} if "cpu_baseline" in d else None)
        if "e2e" in d:
            for k,v in d["e2e"]["runs"].items(): print("  e2e",k,{x:v[x] for x in ("reference_s","ours_s","ours_runs_s","speedup","parity")}); print("     ", v["ours_log"])
    except Exception as e: print(c, "failed", e)
PY
