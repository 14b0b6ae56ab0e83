#!/bin/bash
# round 4: three PMC groups of the C3 fused stream kernel (DMND_SEED_CLASSES from the environment), summary into gpurun_out/c3/
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/c3"; mkdir -p "$OUT"
TMP=$(mktemp -d /tmp/pmc.XXXXXX)
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --config C3 --steps 1 --warmup 1 --no-cpu-baseline --no-pipeline"
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "FETCH_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$TMP/p$i" -o pmc -- $B > "$TMP/p$i.log" 2>&1 || tail -5 "$TMP/p$i.log"
done
python "$ROOT/tools/pmc_summary.py" "$OUT/pmc_C3_classes${DMND_SEED_CLASSES:-1}.json" "$TMP"/p*/
rm -rf "$TMP"
python - "$OUT/pmc_C3_classes${DMND_SEED_CLASSES:-1}.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if "seed_stream_fast" in k: print(k[:80], json.dumps({a: (round(b) if isinstance(b, float) else b) for a, b in v.items()}))
PY
