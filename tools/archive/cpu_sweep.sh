#!/bin/bash
# usage: tools/cpu_sweep.sh "THREADS..." "SPINS..." [env assignments...]: bench.py (40 steps) per (host threads, pool spins) pair
T=${1:-"32"}; S=${2:-"600"}; shift 2
for t in $T; do for s in $S; do
  env DMND_POOL_SPINS=$s "$@" timeout 200 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --host-threads $t > gpurun_out/cs_${t}_${s}.json 2> gpurun_out/cs_${t}_${s}.err
  python - <<PY
import json, statistics
f="gpurun_out/cs_${t}_${s}"
try:
    d=json.loads(open(f+".json").read().strip().splitlines()[-1])
    e=d["ms_each_step"]
    print("threads ${t} spins ${s}", "$@", "GCUPS", round(d["value"],1), "ms/step", round(d["ms_per_step"],2), "median", statistics.median(e), "max", max(e), "n>12ms", sum(x>12 for x in e), "cpu_ms/step", round(d["host_cpu_ms_per_step"],1), "serial cpu", {k: round(v,1) for k,v in d["host_cpu_ms_last_step"].items()})
except Exception as ex:
    print(f, "ERR", ex, open(f+".err").read()[-600:])
PY
done; done
grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat
