set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -3 | cut -c1-300
DMND_BENCH_SHARE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/b_share2.json 2> gpurun_out/b_share2.err; tail -c 600 gpurun_out/b_share2.json | head -c 600; echo
timeout 900 bash tools/profile_round.sh r01g > gpurun_out/profile_r01g.log 2>&1; tail -5 gpurun_out/profile_r01g.log
timeout 400 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_r01g.json 2> gpurun_out/bench_r01g.err; head -c 700 gpurun_out/bench_r01g.json; echo
timeout 400 python bench.py > gpurun_out/bench_r01g_default.json 2> gpurun_out/bench_r01g_default.err; head -c 400 gpurun_out/bench_r01g_default.json; echo
