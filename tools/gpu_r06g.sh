#!/bin/bash
# Round 6: time line of ONE extension call on an idle GPU (the serial steps bench.py appends): kernel trace of a short C2 run
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r06g"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cfg=${1:-C2}
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/tr" -o t -- python "$ROOT/bench.py" --config $cfg --steps 4 --warmup 2 --no-cpu-baseline --no-e2e --no-masked-step > "$OUT/trace.log" 2>&1
find "$OUT/tr" -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_trace_$cfg.csv"
find "$OUT/tr" -name "*memory_copy_trace.csv" | head -1 | xargs -I{} cp {} "$OUT/memcopy_trace_$cfg.csv"
rm -rf "$OUT/tr"
python - "$OUT/kernel_trace_$cfg.csv" "$OUT/memcopy_trace_$cfg.csv" <<'PY'
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Stream_Id", r.get("Queue_Id", ""))))
try:
    for r in csv.DictReader(open(sys.argv[2])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "")), r.get("Stream_Id", "")))
except Exception as e:
    print("no copy trace", e)
ev.sort()
# the last extension call: from the last hauser_bias_kernel (or xdrop_seg_kernel) to the end
last = max(i for i, e in enumerate(ev) if "xdrop_seg_kernel" in e[2])
start = last
while start > 0 and "hauser_bias" not in ev[start][2] and last - start < 4: start -= 1
t0 = ev[start][0]
prev_end = t0
for s, e, name, q in ev[start:]:
    print("%9.1f us  dur %8.1f  gap %7.1f  %s  [%s]" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name, q))
    prev_end = max(prev_end, e)
PY
