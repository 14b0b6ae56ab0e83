#!/bin/bash
# round 4: the smaller items -- RCCL world-size-1 exchange, two contexts with different motif tables, C5 at full size, the bench line with its scaling model
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; OUT="$ROOT/gpurun_out/small"; mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_db_shard.py tests/test_gpu_mask.py -m gpu -x -q -k "rccl or motif_tables" 2>&1 | tail -8
/usr/bin/time -v timeout 1200 python -m pytest tests/test_gpu_bench.py -m gpu -x -q -k "c5_full" 2>&1 | grep -E "passed|failed|Elapsed|Error|assert" | tail -8
timeout 600 python bench.py --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > "$OUT/bench_C2.json"
python -c "
import json; d=json.loads(open('$OUT/bench_C2.json').read()); print(d['ms_per_step'], json.dumps(d.get('scaling_model'))[:900])"
