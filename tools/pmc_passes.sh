#!/bin/bash
# PMC passes of one bench configuration on the GPU box (through gpurun): every counter group in its own rocprofv3 run with
# --kernel-trace only (never combined with another trace domain), aggregated per kernel and launch by tools/pmc_summary.py.
#   tools/pmc_passes.sh C2|C3|C4|C5 OUT.json [extra bench.py arguments]
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
CFG=$1; OUT=$2; shift 2
TMP=$(mktemp -d /tmp/pmc.XXXXXX)
cd /tmp && export TMPDIR=/tmp
STEPS=3; [ "$CFG" = C3 ] && STEPS=1
B="python $ROOT/bench.py --config $CFG --steps $STEPS --warmup 1 --no-cpu-baseline --no-pipeline --no-masked-step $*"
i=0
# PMC_SHORT=1: the four passes the roofline object needs (bytes, L2 requests / misses / fabric reads, instruction counts): the large
# configurations spend a minute per pass on set-up alone
PMC_GROUPS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE")
[ "${PMC_SHORT:-0}" = 1 ] && PMC_GROUPS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES")
for grp in "${PMC_GROUPS[@]}"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$TMP/p$i" -o pmc -- $B > "$TMP/p$i.log" 2>&1 || tail -5 "$TMP/p$i.log"
done
python "$ROOT/tools/pmc_summary.py" "$OUT" "$TMP"/p*/
rm -rf "$TMP"
