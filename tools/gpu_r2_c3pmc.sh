#!/bin/bash
# PMC passes (separate from any trace domain other than kernel-trace) for the short-seed stream kernel
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c3pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
B="python $GRAFT_REPO_ROOT/bench.py --config C3 --steps 1 --warmup 0 --no-cpu-baseline"
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- $B > $OUT/p$i.log 2>&1
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/pmc_summary.json $OUT/p*/ > /dev/null 2>&1
python - <<PY
import json
d = json.load(open("$OUT/pmc_summary.json"))
for k, v in d.items():
    if "stream" in k or "post" in k or "collect" in k:
        print(k[:60]); [print("   ", a, "%.4g" % b) for a, b in sorted(v.items())]
PY
rm -rf $OUT/p*/
