#!/bin/bash
# C2: seed stages in flight (1, 2, 3) x extension contexts (3, 4) -- is the one seed stage at a time what the step waits for?
mkdir -p gpurun_out/r06v
for sc in 1 2 3; do for ec in 3 4; do
  for rep in 1 2; do
  timeout 600 python bench.py --steps 60 --warmup 10 --no-e2e --no-masked-step --no-cpu-baseline --seed-contexts $sc --ext-contexts $ec 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('C2 seed-contexts $sc ext-contexts $ec: ms/step %.3f  seed call p50 %.2f ext p50 %.2f  host cpu %.2f' % (d['ms_per_step'], d['latency_in_pipeline']['seed_stage_call_ms']['p50'], d['latency_in_pipeline']['extension_of_a_batch_ms']['p50'], d['host_cpu_ms_per_step']))
"
  done
done; done
