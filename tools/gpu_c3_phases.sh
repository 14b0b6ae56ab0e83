#!/bin/bash
# round 4: where the workgroups of the C3 fused stream kernel spend their time (DMND_SEED_PHASES), classes on and off
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
for cls in ${CLS:-1 0}; do
  DMND_SEED_PHASES=1 DMND_SEED_CLASSES=$cls timeout 600 python bench.py --config C3 --steps 1 --warmup 1 --no-cpu-baseline --no-pipeline 2> /tmp/err.txt | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('classes $cls ms/step %.1f stream %.1f' % (d['ms_per_step'], d['seed_kernel_ms']['stream_reference']))"
  grep SEED_PHASES /tmp/err.txt | tail -1
done
