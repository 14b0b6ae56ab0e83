#!/bin/bash
# round 3, third GPU call: remaining GPU tests, level-1 filter / probe policy sweep (this time with the library that has the knobs), CLI wall times
set -u
ROOT="$GRAFT_REPO_ROOT"; OUT="$ROOT/gpurun_out/r03c"; mkdir -p "$OUT"
cd "$ROOT"
cat /sys/kernel/mm/transparent_hugepage/enabled; nproc; cat /sys/fs/cgroup/cpu.max
timeout 900 python -m pytest tests/test_gpu_mask.py tests/test_gpu_edge_cases.py tests/test_gpu_cli.py tests/test_gpu_seed.py tests/test_gpu_fullscale.py -m gpu -x -q 2>&1 | tail -8
timeout 600 python tools/stream_sweep.py C2 > "$OUT/stream_sweep_C2.txt" 2>&1; cat "$OUT/stream_sweep_C2.txt"
python - <<PY
import os, sys, subprocess, time, hashlib
sys.path.insert(0, "$ROOT")
import bench
w = bench.Workload("C2", 100_000, 10_000, 1, 0, "db")
d = "/tmp/e2e"; os.makedirs(d, exist_ok=True)
q = w.write_fasta(d)
subprocess.run(["$ROOT/oracle/_ref/diamond", "makedb", "--in", d + "/db.faa", "-d", d + "/db", "-p", "16"], check=True, capture_output=True)
def run(flags, env):
    t0 = time.perf_counter()
    p = subprocess.Popen(["$ROOT/diamond_amd/diamond-hip", "blastp", "--fast", "-q", q, "-d", d + "/db", "-o", d + "/o.tsv"] + flags, stderr=subprocess.PIPE, text=True, env=dict(os.environ, **env))
    t_total = None
    lines = []
    for l in p.stderr:
        lines.append(l)
        if l.startswith("Total time"): t_total = time.perf_counter() - t0
    p.wait()
    return time.perf_counter() - t0, t_total, "".join(lines)
for flags in (["--masking", "0", "--motif-masking", "0", "--algo", "0"], ["--algo", "0"], []):
    for env in ({}, {"DMND_CLI_CLEAN_EXIT": "1"}):
        res = [run(flags, dict(env, DMND_CLI_TIMELINE="1")) for _ in range(4)]
        print("== flags", flags, env, "wall", [round(r[0], 3) for r in res], "'Total time' line seen at", [round(r[1], 3) for r in res])
    print(res[-1][2][-1800:])
for flags in (["--masking", "0", "--motif-masking", "0", "--algo", "0"], ["--algo", "0"], []):
    t0 = time.perf_counter()
    subprocess.run(["$ROOT/oracle/_ref/diamond", "blastp", "--fast", "-q", q, "-d", d + "/db", "-o", d + "/r.tsv", "-p", "16"] + flags, capture_output=True)
    print("reference", flags, round(time.perf_counter() - t0, 3))
PY
