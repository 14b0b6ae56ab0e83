#!/bin/bash
# Round 6: the one long step of the FIRST bench run on a fresh box (step ~38 of 50: 40 - 50 ms): which phase of which call
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r06o"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
DMND_TRACE=1 DMND_TRACE_ALLOC=1 timeout 600 python "$ROOT/bench.py" --no-cpu-baseline --no-e2e --no-masked-step > "$OUT/first.json" 2> "$OUT/first.err"
python - "$OUT/first.err" "$OUT/first.json" <<'PY'
import re, sys, json
lines = open(sys.argv[1]).read().splitlines()
for i, l in enumerate(lines):
    m = re.match(r"dmnd_extend total ([0-9.]+) ms", l)
    if m and float(m.group(1)) > 15:
        print("---- slow call, line", i)
        print("\n".join(lines[max(0, i - 40):i + 1]))
for l in lines:
    if "hipMalloc" in l or "HostMalloc" in l:
        pass
print("allocs:", sum(1 for l in lines if l.startswith("hipMalloc")))
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1]); e = d["ms_each_step"]
print("ms/step", d["ms_per_step"], "longest", sorted(e)[-3:], "at", e.index(max(e)))
PY
