#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r05k"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  env MODES_EXTEND=0 DMND_SEED_SJ=1 DMND_SEED_SLOTS_X8=${SLOTS:-16} timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/p$i" -o pmc -- python "$ROOT/tools/seed_modes.py" sensitive 1 > "$OUT/p$i.log" 2>&1 || tail -3 "$OUT/p$i.log"
done
python "$ROOT/tools/pmc_summary.py" "$OUT/pmc_sj.json" "$OUT"/p*/ ; rm -rf "$OUT"/p[0-9]
python - <<PY
import json
d=json.load(open("$OUT/pmc_sj.json"))
for k,v in d.items():
    if "seed_sj" in k: print(k[:40], {a:round(b/1e6,2) if isinstance(b,float) else b for a,b in v.items()})
PY
