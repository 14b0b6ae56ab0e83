#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2busy
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 5 --no-cpu-baseline > $OUT/bench.log 2>&1
f=$(find $OUT/tr -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/gpu_busy.py $f $OUT/gpu_busy_C2.json 0.25
tail -1 $OUT/bench.log | cut -c1-200
rm -rf $OUT/tr
