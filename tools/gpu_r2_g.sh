#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
for p in 0 1 2; do
  DMND_SEED_PROBE=$p timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('probe $p', d['seed_kernel_ms'], 'ms/step', round(d['ms_per_step'],2), 'hits', d['config']['workload'][-90:])"
done
