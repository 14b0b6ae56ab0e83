#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for E in 1 2 3; do
  timeout 120 python tools/pipe_probe.py $E 60 8 2>&1 | tail -1
  DMND_EXTEND_TEAM=16 timeout 120 python tools/pipe_probe.py $E 60 16 2>&1 | tail -1
done
done
DMND_EXTEND_TEAM=12 timeout 120 python tools/pipe_probe.py 1 60 12 2>&1 | tail -1
DMND_EXTEND_TEAM=12 timeout 120 python tools/pipe_probe.py 2 60 12 2>&1 | tail -1
