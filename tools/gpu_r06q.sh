#!/bin/bash
# row classes of the packed 16-bit sweeps: the sweep tests, the extension tests, then the bench lines that show the sweeps
mkdir -p gpurun_out/r06q
timeout 1500 python -m pytest tests/test_gpu_swipe.py tests/test_gpu_extend.py tests/test_gpu_extend_device.py -x -q > gpurun_out/r06q/tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r06q/tests.log
for cfg in C2 C2skew; do
  timeout 900 python bench.py --config $cfg --steps 20 --warmup 5 --no-e2e --no-masked-step > gpurun_out/r06q/$cfg.log 2>&1; echo "$cfg rc=$?"
  python - <<PY
import json
for l in open("gpurun_out/r06q/$cfg.log"):
    if l.startswith("{"):
        d=json.loads(l); print("$cfg", d["summary"]); print(d["extension"]); print(d["sweep_roofline"].get("live"))
PY
  DMND_SWEEP_ROWS=0 timeout 900 python bench.py --config $cfg --steps 20 --warmup 5 --no-e2e --no-masked-step --no-cpu-baseline > gpurun_out/r06q/${cfg}_off.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/r06q/${cfg}_off.log"):
    if l.startswith("{"):
        d=json.loads(l); print("$cfg rows off", d["summary"]); print(d["extension"])
PY
done
