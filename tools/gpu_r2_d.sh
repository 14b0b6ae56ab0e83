#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4; do
  DMND_EXTEND_SPLIT=4 DMND_EXTEND_RUNNERS=4 timeout 120 python tools/pipe_probe.py 2 40 8 2>&1 | tail -1
  DMND_SWIPE32=1 DMND_EXTEND_SPLIT=4 DMND_EXTEND_RUNNERS=4 timeout 120 python tools/pipe_probe.py 2 40 8 2>&1 | tail -1
done
for rep in 1 2; do
  timeout 120 python tools/pipe_probe.py 3 40 8 2>&1 | tail -1
  DMND_SWIPE32=1 timeout 120 python tools/pipe_probe.py 3 40 8 2>&1 | tail -1
done
