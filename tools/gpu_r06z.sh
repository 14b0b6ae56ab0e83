#!/bin/bash
# eight ranks on one GPU (gloo exchange), no --shard: the default decomposition at N = 8 (the committed model's best: 2 x N/2 for C2)
mkdir -p gpurun_out/r06z
DMND_BENCH_SHARE_GPU=1 timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29547 bench.py --gpus 8 --steps 6 --warmup 2 --no-e2e --no-masked-step > gpurun_out/r06z/n8.log 2>&1; echo "n8 rc=$?"
python - <<PY
import json
for l in open("gpurun_out/r06z/n8.log"):
    if l.startswith("{"):
        d=json.loads(l); print(d["summary"]["ms_per_step"], d["parity_checked"], d["parity"]["note"], d["rccl"]["shard"], d["rccl"]["world_size"], d["config"]["parallelism"][:50])
PY
grep -i "error\|Traceback" gpurun_out/r06z/n8.log | head -5
