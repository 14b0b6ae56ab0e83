#!/bin/bash
# usage: tools/split_sweep.sh "RUNNERS..." "SPLITS..." [extra bench args]: bench.py for every (runners, sub-batches) pair of dmnd_extend
R=${1:-"3 4"}; S=${2:-"8 12"}; shift 2
for r in $R; do for s in $S; do
  DMND_EXTEND_RUNNERS=$r DMND_EXTEND_SPLIT=$s timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/sw_${r}_${s}.json 2> gpurun_out/sw_${r}_${s}.err
  python - <<PY
import json
f="gpurun_out/sw_${r}_${s}"
try:
    d=json.loads(open(f+".json").read().strip().splitlines()[-1])
    print("runners ${r} split ${s}", "$@", round(d["value"],1), round(d["ms_per_step"],2), round(d["roofline"]["kernel_ms"],3), {k:round(v,2) for k,v in d["wall_ms_last_step"].items()}, {k:round(v,2) for k,v in (d.get("pipeline_wall_ms_per_step") or {}).items()}, {k:round(v,2) for k,v in d["extension"].items() if k.endswith("ms")}, "PIPE", {k:round(v,2) for k,v in (d.get("pipeline_extension_last_step") or {}).items() if k.endswith("ms")})
except Exception as e:
    print(f, "ERR", e, open(f+".err").read()[-600:])
PY
done; done
