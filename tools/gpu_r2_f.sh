#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2f_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2f_pytest.log
for E in 1 2 3; do
  DMND_EXTEND_SPLIT=1 DMND_EXTEND_RUNNERS=1 timeout 120 python tools/pipe_probe.py $E 40 8 2>&1 | tail -1
  DMND_EXTEND_SPLIT=1 DMND_EXTEND_RUNNERS=1 timeout 120 python tools/pipe_probe.py $E 40 16 2>&1 | tail -1
  timeout 120 python tools/pipe_probe.py $E 40 8 2>&1 | tail -1
done
DMND_TRACE=1 DMND_EXTEND_SPLIT=1 DMND_EXTEND_RUNNERS=1 timeout 120 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-pipeline --host-threads 8 2>&1 | grep -E "dmnd_extend" | tail -4
