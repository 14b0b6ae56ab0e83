#!/bin/bash
# the row classes kept for the later iterations of a call that began in them: C2skew (and C3) lines, extension tests
mkdir -p gpurun_out/r06ad
timeout 1200 python -m pytest tests/test_gpu_extend_device.py tests/test_gpu_skew.py -x -q > gpurun_out/r06ad/t.log 2>&1; tail -2 gpurun_out/r06ad/t.log
for cfg in C2skew C3; do
  steps=40; [ $cfg = C3 ] && steps=12
  for rep in 1 2; do
  timeout 900 python bench.py --config $cfg --steps $steps --warmup 5 --no-e2e --no-masked-step --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); e=d['extension']; print('$cfg ms/step %.3f lane use %.3f sweeps r1 %.2f r2 %.2f tb %.2f' % (d['ms_per_step'], d['sweep_roofline']['lane_use'], e['round1_swipe_kernel_ms'], e['round2_swipe_kernel_ms'], e['traceback_kernel_ms']))
"
  done
done
