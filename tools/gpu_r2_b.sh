#!/bin/bash
# round 2: kernel stats of the 16-bit vs 32-bit sweep, and how the extension call reacts to the runner split
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
for k in 16 32; do
  if [ $k = 32 ]; then export DMND_SWIPE32=1; else unset DMND_SWIPE32; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats$k -o s$k -- $B > $OUT/stats$k.log 2>&1
  find $OUT/stats$k -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$k.csv
  echo "== kernel stats k$k"; head -14 $OUT/kernel_stats_$k.csv | cut -c1-200
done
unset DMND_SWIPE32
for cfg in "1 1" "2 2" "4 4" "8 8" "16 8"; do
  set -- $cfg
  for k in 16 32; do
    if [ $k = 32 ]; then export DMND_SWIPE32=1; else unset DMND_SWIPE32; fi
    DMND_EXTEND_SPLIT=$1 DMND_EXTEND_RUNNERS=$2 timeout 200 $B --no-pipeline > $OUT/b_$1_$2_$k.json 2>/dev/null
    python - $OUT/b_$1_$2_$k.json "split=$1 runners=$2 k$k" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    e = d["extension"]
    print(sys.argv[2], "ms/step %.2f" % d["ms_per_step"], "wall", {k: round(v, 2) for k, v in d["wall_ms_last_step"].items()}, "sweep_ms %.2f walk_ms %.2f" % (e["round1_swipe_kernel_ms"], e["traceback_kernel_ms"]), "cpu %.1f" % d["host_cpu_ms_per_step"])
except Exception as ex:
    print(sys.argv[2], "failed", ex)
PY
  done
done
