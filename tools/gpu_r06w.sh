#!/bin/bash
# seed contexts 2 / extension contexts 4 against the defaults (1 / 3) on the other single-block configs
mkdir -p gpurun_out/r06w
for cfg in C2skew C4 C3; do
  steps=40; [ $cfg = C3 ] && steps=12
  for set in "" "--seed-contexts 2 --ext-contexts 4"; do
  for rep in 1 2; do
  timeout 900 python bench.py --config $cfg --steps $steps --warmup 5 --no-e2e --no-masked-step --no-cpu-baseline $set 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$cfg [$set]: ms/step %.3f  seed call p50 %.2f ext p50 %.2f  host cpu %.2f' % (d['ms_per_step'], d['latency_in_pipeline']['seed_stage_call_ms']['p50'], d['latency_in_pipeline']['extension_of_a_batch_ms']['p50'], d['host_cpu_ms_per_step']))
"
  done; done
done
