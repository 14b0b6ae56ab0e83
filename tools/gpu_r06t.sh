#!/bin/bash
# A/B of one switch of the sweeps on the bench lines: tools/gpu_r06t.sh ENVVAR   (runs each config with ENVVAR unset and =0, twice)
mkdir -p gpurun_out/r06t
V=$1
for rep in 1 2; do
for cfg in C2 C2skew C3; do
  steps=40; [ $cfg = C3 ] && steps=12
  for off in "" 0; do
    if [ -n "$off" ]; then export $V=0; else unset $V; fi
    timeout 900 python bench.py --config $cfg --steps $steps --warmup 5 --no-e2e --no-masked-step --no-cpu-baseline > gpurun_out/r06t/$cfg.log 2>&1
    python - <<PY
import json
for l in open("gpurun_out/r06t/$cfg.log"):
    if l.startswith("{"):
        d=json.loads(l); e=d["extension"]; print("$cfg $V=${off:-default} ms/step %.3f  sweeps r1 %.3f r2 %.3f tb %.3f" % (d["ms_per_step"], e["round1_swipe_kernel_ms"], e["round2_swipe_kernel_ms"], e["traceback_kernel_ms"]))
PY
  done
done
done
