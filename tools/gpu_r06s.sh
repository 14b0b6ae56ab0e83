#!/bin/bash
# row classes from 32768 items on: the sweep / extension suites, C2 (must not change) and C2skew (rows), C3 (rows)
mkdir -p gpurun_out/r06s
timeout 1500 python -m pytest tests/test_gpu_swipe.py tests/test_gpu_extend.py tests/test_gpu_extend_device.py tests/test_gpu_skew.py -x -q > gpurun_out/r06s/tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r06s/tests.log
for cfg in C2 C2skew C3; do
  steps=30; [ $cfg = C3 ] && steps=12
  timeout 900 python bench.py --config $cfg --steps $steps --warmup 5 --no-e2e --no-masked-step > gpurun_out/r06s/$cfg.log 2>&1; echo "$cfg rc=$?"
  python - <<PY
import json
for l in open("gpurun_out/r06s/$cfg.log"):
    if l.startswith("{"):
        d=json.loads(l); print("$cfg", d["summary"]); print({k:v for k,v in d["extension"].items() if "kernel" in k or "cells" in k})
PY
done
