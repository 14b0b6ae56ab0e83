#!/bin/bash
# round 4: kernel-by-kernel timeline of ONE C2 seed stage (kernel trace incl. the runtime's fill / copy kernels), for the step <= 2.8 ms work
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; OUT="$ROOT/gpurun_out/c2trace"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d "$OUT/tr" -o t -- python "$ROOT/bench.py" --steps 6 --warmup 3 --no-cpu-baseline --no-pipeline > "$OUT/bench.json" 2> "$OUT/bench.err"
f=$(find "$OUT/tr" -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee "$OUT/step_timeline.txt"
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last seed stage: from the last seed_qid / seed_index kernel group backwards -- take the last 'seed_index_kernel' pair (2 shapes)
idx = [i for i, r in enumerate(rows) if "seed_index_kernel" in r["Kernel_Name"]]
start = idx[-2]
while start > 0 and int(rows[start]["Start_Timestamp"]) - int(rows[start - 1]["End_Timestamp"]) < 200000 and "traceback" not in rows[start - 1]["Kernel_Name"]: start -= 1
t0 = int(rows[start]["Start_Timestamp"]); prev_end = t0
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("dmnd::", "").replace("rocprim::ROCPRIM_400200_NS::detail::", "rp::")[:70]
    print("%9.1f us  +%7.1f gap  %8.1f us  q%s  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), name))
    prev_end = max(prev_end, e)
PY
rm -rf "$OUT/tr"
python -c "
import json
d = json.loads([l for l in open('$OUT/bench.json') if l.startswith('{')][0])
print('ms_per_step', d['ms_per_step'], d.get('seed_kernel_ms'))"
