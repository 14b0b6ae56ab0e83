#!/bin/bash
# four ranks on one GPU (gloo exchange): two query halves x two database shards against the database-sharded run -- parity at N = 4
mkdir -p gpurun_out/r06y
for shard in 2d db; do
DMND_BENCH_SHARE_GPU=1 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 4 --steps 8 --warmup 2 --no-e2e --no-masked-step --shard $shard > gpurun_out/r06y/n4_$shard.log 2>&1; echo "n4 $shard rc=$?"
python - <<PY
import json
for l in open("gpurun_out/r06y/n4_$shard.log"):
    if l.startswith("{"):
        d=json.loads(l); print("$shard", d["summary"]["ms_per_step"], d["parity_checked"], d["parity"], d["rccl"], d["config"]["parallelism"][:60])
PY
tail -3 gpurun_out/r06y/n4_$shard.log | cut -c1-300 | grep -v "^{" 
done
