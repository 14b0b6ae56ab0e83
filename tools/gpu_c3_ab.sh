#!/bin/bash
# round 4: C3 (--sensitive) with key classes on / off (DMND_SEED_CLASSES): parity at reduced size, then the bench step and the stream kernel's time
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/c3"; mkdir -p "$OUT"
cd "$ROOT"
[ -n "$SKIP_TESTS" ] || timeout 900 python -m pytest tests/test_gpu_seed.py tests/test_gpu_bench.py -m gpu -x -q -k "not two_ranks and not c5_full and not e2e" 2>&1 | tail -4
for cls in ${CLS:-1 0}; do
  DMND_SEED_CLASSES=$cls timeout 600 python bench.py --config C3 --steps 4 --warmup 2 --no-e2e > "$OUT/bench_C3_classes$cls.json" 2> "$OUT/err$cls.txt"
  python - "$OUT/bench_C3_classes$cls.json" $cls <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("classes", sys.argv[2], "ms/step %.2f" % d["ms_per_step"], "parity", d.get("parity_checked"), "seed_kernel_ms", {k: round(v, 2) for k, v in d["seed_kernel_ms"].items()}, "roofline frac", round(d["roofline"]["frac"], 4))
PY
done
