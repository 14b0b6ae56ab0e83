#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_extend.py tests/test_gpu_cli.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -2
for E in 1 2 3; do
  timeout 120 python tools/pipe_probe.py $E 40 8 2>&1 | tail -1
  DMND_EXTEND_TEAM=16 timeout 120 python tools/pipe_probe.py $E 40 16 2>&1 | tail -1
  DMND_EXTEND_TEAM=4 timeout 120 python tools/pipe_probe.py $E 40 4 2>&1 | tail -1
done
DMND_EXTEND_SPLIT=4 DMND_EXTEND_RUNNERS=4 timeout 120 python tools/pipe_probe.py 2 40 8 2>&1 | tail -1
DMND_TRACE=1 timeout 120 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-pipeline --host-threads 8 2>&1 | grep -E "dmnd_extend" | tail -2
