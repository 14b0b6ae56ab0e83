#!/bin/bash
# round 4, last check of the tree as committed: masking tests, a slice of the CLI A/B tests, smoke(), the C2 bench line with e2e
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; OUT="$ROOT/gpurun_out/r04final"; mkdir -p "$OUT"
timeout 120 python -m pytest tests/test_gpu_mask.py tests/test_gpu_seed.py -m gpu -x -q 2>&1 | tail -2
timeout 240 python -m pytest tests/test_gpu_cli.py -m gpu -x -q -k "default_masking or query_indexed or makedb_blastp or frameshift or tiny_inputs or gpus_distributes" 2>&1 | tail -2
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 400 python bench.py --with-masking > "$OUT/bench_C2_final.json" 2> "$OUT/bench_C2_final.err"
python - "$OUT/bench_C2_final.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("C2 ms/step %.3f value %.1f parity %s roofline %.3f l2 %.3f" % (d["ms_per_step"], d["value"], d.get("parity_checked"), d["roofline"]["frac"], d["roofline"].get("l2_requests_frac", -1)))
print({k: (round(v["speedup"], 2), v["ours_runs_s"], v["parity"]) for k, v in d["e2e"]["runs"].items()})
print("masked_step", d["masked_step"]["ms_per_step"], d["masked_step"]["parts_ms"], d["masked_step"]["parity"]["matches"])
PY
