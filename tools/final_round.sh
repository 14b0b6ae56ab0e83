#!/bin/bash
# Runs on the GPU box at the end of a round: full GPU suite, 2-rank code path on one GPU, rocprofv3 passes, bench lines.
# usage: tools/final_round.sh TAG
set -u
TAG="${1:-r01}"
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -2 | cut -c1-200
DMND_BENCH_SHARE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${TAG}_share2.json 2> gpurun_out/bench_${TAG}_share2.err; head -c 300 gpurun_out/bench_${TAG}_share2.json; echo
timeout 900 bash tools/profile_round.sh $TAG > gpurun_out/profile_$TAG.log 2>&1; tail -3 gpurun_out/profile_$TAG.log
cp gpurun_out/$TAG/pmc_summary.json profiles/r01_pmc_summary.json
timeout 400 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; head -c 300 gpurun_out/bench_$TAG.json; echo
timeout 400 python bench.py > gpurun_out/bench_${TAG}_default.json 2> gpurun_out/bench_${TAG}_default.err; head -c 300 gpurun_out/bench_${TAG}_default.json; echo
timeout 200 python bench.py --no-pipeline --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/bench_${TAG}_serial.json 2> gpurun_out/bench_${TAG}_serial.err; head -c 300 gpurun_out/bench_${TAG}_serial.json; echo
