#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
for T in 8 16; do
echo "== split 1, threads $T"
DMND_TRACE=1 DMND_EXTEND_SPLIT=1 DMND_EXTEND_RUNNERS=1 timeout 120 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-pipeline --host-threads $T 2>&1 | grep -E "dmnd_extend" | tail -4
done
echo "== default"
DMND_TRACE=1 timeout 120 python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-pipeline --host-threads 8 2>&1 | grep -E "dmnd_extend" | tail -10
