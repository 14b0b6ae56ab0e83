#!/bin/bash
# seed_leftmost_kernel as a bounded grid: seed tests, then C3 / C2 / C2skew lines
mkdir -p gpurun_out/r06aa
timeout 1200 python -m pytest tests/test_gpu_seed.py tests/test_gpu_extend.py -x -q > gpurun_out/r06aa/t.log 2>&1; tail -2 gpurun_out/r06aa/t.log
for cfg in C3 C2 C2skew; do
  steps=40; [ $cfg = C3 ] && steps=12
  for rep in 1 2; do
  timeout 900 python bench.py --config $cfg --steps $steps --warmup 5 --no-e2e --no-masked-step --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$cfg ms/step %.3f  seed call p50 %.2f  seed kernels %s' % (d['ms_per_step'], d['latency_in_pipeline']['seed_stage_call_ms']['p50'], {k:round(v,2) for k,v in d['seed_kernel_ms'].items()}))
"
  done
done
