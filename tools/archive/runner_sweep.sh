#!/bin/bash
# usage: tools/runner_sweep.sh "R:T ..."  -- bench.py with R extension runners (= sub-batches) and T host threads
for rt in $1; do r=${rt%%:*}; t=${rt##*:}
  DMND_EXTEND_RUNNERS=$r DMND_EXTEND_SPLIT=$r timeout 200 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --host-threads $t ${2:-} > gpurun_out/rs_${r}_${t}.json 2> gpurun_out/rs_${r}_${t}.err
  python - <<PY
import json, statistics
f="gpurun_out/rs_${r}_${t}"
try:
    d=json.loads(open(f+".json").read().strip().splitlines()[-1])
    e=d.get("ms_each_step") or [0]
    print("runners ${r} threads ${t} ${2:-}", "GCUPS", round(d["value"],1), "ms/step", round(d["ms_per_step"],2), "median", statistics.median(e), "max", max(e), "cpu_ms/step", round(d["host_cpu_ms_per_step"],1), "serial wall", {k: round(v,2) for k,v in d["wall_ms_last_step"].items()}, "serial cpu", {k: round(v,1) for k,v in d["host_cpu_ms_last_step"].items()})
except Exception as ex:
    print(f, "ERR", ex, open(f+".err").read()[-600:])
PY
done
grep -E "nr_throttled" /sys/fs/cgroup/cpu.stat
