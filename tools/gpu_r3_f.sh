#!/bin/bash
# round 3, sixth GPU call: concurrent seed stages in the bench pipeline, C5 host thread settings, CLI timeline detail
set -u
ROOT="$GRAFT_REPO_ROOT"; OUT="$ROOT/gpurun_out/r03f"; mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python -m pytest tests/test_gpu_seed.py tests/test_gpu_bench.py -m gpu -x -q 2>&1 | tail -4
for sc in 1 2 3; do
  timeout 300 python bench.py --config C2 --steps 30 --warmup 6 --no-cpu-baseline --seed-contexts $sc > "$OUT/bench_C2_sc$sc.json" 2> "$OUT/err"; tail -c 200 "$OUT/err"
done
timeout 300 python bench.py --config C2 --steps 30 --warmup 6 --no-cpu-baseline --seed-contexts 2 --ext-contexts 4 --host-threads 16 > "$OUT/bench_C2_sc2_e4.json" 2> "$OUT/err"
timeout 300 python bench.py --config C4 --steps 10 --warmup 3 --no-cpu-baseline --seed-contexts 2 > "$OUT/bench_C4_sc2.json" 2> "$OUT/err"
timeout 300 python bench.py --config C4 --steps 10 --warmup 3 --no-cpu-baseline --seed-contexts 1 > "$OUT/bench_C4_sc1.json" 2> "$OUT/err"
timeout 300 python bench.py --config C3 --steps 4 --warmup 1 --no-cpu-baseline --seed-contexts 2 > "$OUT/bench_C3_sc2.json" 2> "$OUT/err"
timeout 400 python bench.py --config C5 --steps 4 --warmup 2 --no-cpu-baseline --host-threads 16 --ext-contexts 4 > "$OUT/bench_C5_t16e4.json" 2> "$OUT/err"; tail -c 200 "$OUT/err"
timeout 400 python bench.py --config C5 --steps 4 --warmup 2 --no-cpu-baseline --host-threads 16 --ext-contexts 2 > "$OUT/bench_C5_t16e2.json" 2> "$OUT/err"; tail -c 200 "$OUT/err"
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "ms/step %.3f value %.1f cpu_ms %.1f stream_ms %.3f seed_call p50 %.2f ext p50 %.2f" % (d["ms_per_step"], d["value"], d["host_cpu_ms_per_step"], d["roofline"]["kernel_ms"], d["latency_in_pipeline"]["seed_stage_call_ms"]["p50"], d["latency_in_pipeline"]["extension_of_a_batch_ms"]["p50"]))
    except Exception as e: print(f, "failed", e)
PY
python - <<PY
import os, sys, subprocess, time
sys.path.insert(0, "$ROOT")
import bench
w = bench.Workload("C2", 100_000, 10_000, 1, 0, "db")
d = "/tmp/e2e"; os.makedirs(d, exist_ok=True)
q = w.write_fasta(d)
subprocess.run(["$ROOT/oracle/_ref/diamond", "makedb", "--in", d + "/db.faa", "-d", d + "/db", "-p", "16"], check=True, capture_output=True)
def run(flags, env):
    time.sleep(1.0)
    t0 = time.perf_counter()
    p = subprocess.run(["$ROOT/diamond_amd/diamond-hip", "blastp", "--fast", "-q", q, "-d", d + "/db", "-o", d + "/o.tsv"] + flags, capture_output=True, text=True, env=dict(os.environ, **env))
    return time.perf_counter() - t0, p.stderr
for flags in (["--masking", "0", "--motif-masking", "0", "--algo", "0"], ["--algo", "0"], []):
    res = [run(flags, {"DMND_CLI_TIMELINE": "1"}) for _ in range(4)]
    print("== flags", flags, "wall (1 s apart)", [round(r[0], 3) for r in res])
    print("\n".join(l for l in res[-1][1].splitlines() if l.startswith(("timeline", "Total"))))
PY
