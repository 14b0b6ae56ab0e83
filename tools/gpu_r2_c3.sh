#!/bin/bash
# kernel-level picture of the seed stage with short seeds (C3: --sensitive)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --config C3 --steps 2 --warmup 1 --no-cpu-baseline"
DMND_TRACE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $B > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_c3.csv
head -30 $OUT/kernel_stats_c3.csv | cut -c1-220
grep "dmnd_seed_search" $OUT/stats.log | tail -2
tail -1 $OUT/stats.log | cut -c1-600
rm -rf $OUT/stats
