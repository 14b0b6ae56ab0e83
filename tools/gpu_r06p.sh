#!/bin/bash
# bench tail after the reorder: N=1 default line and the 2-rank shared-GPU run (parity at N>1 behind the teardown)
mkdir -p gpurun_out/r06p
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r06p/n1.log 2>&1; echo "n1 rc=$?"
tail -c 600 gpurun_out/r06p/n1.log
DMND_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --no-e2e --no-masked-step > gpurun_out/r06p/n2.log 2>&1; echo "n2 rc=$?"
tail -c 700 gpurun_out/r06p/n2.log
