#!/bin/bash
# C5 (8 blocks per batch): seed stages in flight x extension contexts around the defaults (2 / 4); then the bench test
mkdir -p gpurun_out/r06x
for set in "" "--seed-contexts 3" "--ext-contexts 6" "--seed-contexts 3 --ext-contexts 6"; do
  timeout 900 python bench.py --config C5 --steps 24 --warmup 6 --no-e2e --no-masked-step --no-cpu-baseline $set 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('C5 [$set]: ms/step %.3f  seed call p50 %.2f ext p50 %.2f  host cpu %.2f' % (d['ms_per_step'], d['latency_in_pipeline']['seed_stage_call_ms']['p50'], d['latency_in_pipeline']['extension_of_a_batch_ms']['p50'], d['host_cpu_ms_per_step']))
"
done
timeout 900 python -m pytest tests/test_gpu_bench.py -x -q 2>&1 | tail -3
