#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
for E in 1 2 3; do
  timeout 120 python tools/pipe_probe.py $E 30 8 2>&1 | tail -1
  DMND_EXTEND_SPLIT=1 DMND_EXTEND_RUNNERS=1 timeout 120 python tools/pipe_probe.py $E 30 8 2>&1 | tail -1
  DMND_EXTEND_SPLIT=1 DMND_EXTEND_RUNNERS=1 timeout 120 python tools/pipe_probe.py $E 30 4 2>&1 | tail -1
  DMND_EXTEND_SPLIT=4 DMND_EXTEND_RUNNERS=4 timeout 120 python tools/pipe_probe.py $E 30 8 2>&1 | tail -1
done
DMND_SWIPE32=1 timeout 120 python tools/pipe_probe.py 3 30 8 2>&1 | tail -1
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
