#!/bin/bash
# round 3, second GPU call: trace re-layout (swipe tests, bench, walk counters), CLI timeline, level-1 filter sweep
set -u
ROOT="$GRAFT_REPO_ROOT"; OUT="$ROOT/gpurun_out/r03b"; mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_swipe.py tests/test_gpu_extend.py tests/test_gpu_mask.py tests/test_gpu_edge_cases.py tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -8
timeout 300 python tools/stream_sweep.py C2 > "$OUT/stream_sweep_C2.txt" 2>&1; cat "$OUT/stream_sweep_C2.txt"
timeout 600 python bench.py --config C2 --steps 20 --warmup 5 --no-e2e > "$OUT/bench_C2.json" 2> "$OUT/bench_C2.err"; tail -c 300 "$OUT/bench_C2.err"
# CLI timeline on C2 files
python - <<PY
import os, sys, subprocess, time, hashlib
sys.path.insert(0, "$ROOT")
import bench
w = bench.Workload("C2", 100_000, 10_000, 1, 0, "db")
d = "/tmp/e2e"; os.makedirs(d, exist_ok=True)
q = w.write_fasta(d)
subprocess.run(["$ROOT/oracle/_ref/diamond", "makedb", "--in", d + "/db.faa", "-d", d + "/db", "-p", "16"], check=True, capture_output=True)
for flags in (["--masking", "0", "--motif-masking", "0", "--algo", "0"], ["--algo", "0"], []):
    for rep in range(3):
        t0 = time.perf_counter()
        r = subprocess.run(["$ROOT/diamond_amd/diamond-hip", "blastp", "--fast", "-q", q, "-d", d + "/db", "-o", d + "/o.tsv"] + flags, capture_output=True, text=True, env=dict(os.environ, DMND_CLI_TIMELINE="1"))
        wall = time.perf_counter() - t0
    print("== flags", flags, "wall %.3f s rc %d md5 %s" % (wall, r.returncode, hashlib.md5(open(d + "/o.tsv", "rb").read()).hexdigest()))
    print(r.stderr[-3500:])
PY
timeout 400 tools/pmc_passes.sh C2 "$OUT/pmc_summary_C2_trace.json" 2>&1 | tail -2
python - <<PY
import json
d=json.loads(open("$OUT/bench_C2.json").read().strip().splitlines()[-1])
print("C2 ms/step", d["ms_per_step"], "value", d["value"], "parity", d.get("parity_checked"), "ext", d["extension"])
p=json.load(open("$OUT/pmc_summary_C2_trace.json"))
for k,v in p.items():
    if "traceback" in k or "swipe16" in k: print(k[:60], {a:round(b/1e6,1) for a,b in v.items() if "SIZE" in a}, v.get("SQ_INSTS_VALU_per_launch"))
PY
