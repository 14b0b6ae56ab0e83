#!/bin/bash
# round 3, fourth GPU call: the whole GPU suite on the current tree (new trace layout, 3 MB / 3-bit level-1 filter, both bridge seams), C2 counters
set -u
ROOT="$GRAFT_REPO_ROOT"; OUT="$ROOT/gpurun_out/r03d"; mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25
timeout 600 python bench.py --config C2 --steps 20 --warmup 5 > "$OUT/bench_C2.json" 2> "$OUT/bench_C2.err"; tail -c 300 "$OUT/bench_C2.err"
timeout 400 tools/pmc_passes.sh C2 "$OUT/pmc_summary_C2.json" 2>&1 | tail -2
python - <<PY
import json
d=json.loads(open("$OUT/bench_C2.json").read().strip().splitlines()[-1])
print("C2 ms/step", d["ms_per_step"], "value", d["value"], "parity", d.get("parity_checked"), "seed", d["seed_kernel_ms"])
print("e2e", {k:(v["reference_s"], v["ours_s"], v["speedup"], v["parity"]) for k,v in d["e2e"]["runs"].items()})
p=json.load(open("$OUT/pmc_summary_C2.json"))
for k,v in p.items():
    if "stream" in k: print(k[:60], {a:(round(b/1e6,1) if "SIZE" in a else b) for a,b in v.items()})
PY
