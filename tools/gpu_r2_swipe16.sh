#!/bin/bash
# round 2, first GPU check of the packed-int16 two-items-per-wavefront sweep: parity tests, then A/B bench + kernel stats
set -u
OUT=gpurun_out/r2a
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_swipe.py tests/test_gpu_extend.py tests/test_gpu_edge_cases.py -x -q > $OUT/pytest_swipe.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_swipe.log
tail -5 $OUT/pytest_swipe.log
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_k16.json 2> $OUT/bench_k16.err
DMND_SWIPE32=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_k32.json 2> $OUT/bench_k32.err
python - <<'PY'
import json
for k in ("k16", "k32"):
    try:
        d = json.loads(open("gpurun_out/r2a/bench_%s.json" % k).read().strip().splitlines()[-1])
        print(k, "ms/step", round(d["ms_per_step"], 3), "swipe_gcups", d["swipe_kernel_gcups"], "ext", {x: round(v, 3) for x, v in d["extension"].items() if "ms" in x})
    except Exception as e:
        print(k, "failed", e)
PY
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof16 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof16.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT/prof16 -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {}'
