#!/bin/bash
# round 3, fifth GPU call: 64-byte slots with the folded query window (C3), where the stream's L2 misses come from (C2), CLI start-up
set -u
ROOT="$GRAFT_REPO_ROOT"; OUT="$ROOT/gpurun_out/r03e"; mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_seed.py tests/test_gpu_extend.py tests/test_gpu_cli.py "tests/test_gpu_bench.py::test_bench_line_and_parity_at_reduced_size" -m gpu -x -q 2>&1 | tail -6
for wide in 1 0; do
  DMND_SEED_WIDE_SLOTS=$wide timeout 600 python bench.py --config C3 --steps 4 --warmup 1 --no-e2e > "$OUT/bench_C3_wide$wide.json" 2> "$OUT/bench_C3_wide$wide.err"; tail -c 300 "$OUT/bench_C3_wide$wide.err"
done
python - <<PY
import json
for wide in (1, 0):
    d=json.loads(open("$OUT/bench_C3_wide%d.json" % wide).read().strip().splitlines()[-1])
    print("C3 wide", wide, "ms/step", d["ms_per_step"], "parity", d.get("parity_checked"), "seed", d["seed_kernel_ms"], "alone", d["alone"]["seed_kernel_ms"], "cpu_hot_s", d["cpu_baseline"]["hot_path"]["seconds"])
PY
timeout 900 tools/pmc_passes.sh C3 "$OUT/pmc_summary_C3.json" 2>&1 | tail -2
# where the C2 stream kernel's L2 misses come from: variants under one counter group
cd /tmp && export TMPDIR=/tmp
i=0
for v in "X=0" "DMND_SEED_STREAM_NT=1" "DMND_SEED_BM1_KB=2048 DMND_SEED_BM1_K=2" "DMND_SEED_BM1_KB=1024 DMND_SEED_BM1_K=2" "DMND_SEED_LEVEL2=0"; do
  i=$((i+1))
  env $v timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d /tmp/v$i -o pmc -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline > /tmp/v$i.log 2>&1
  python $ROOT/tools/pmc_summary.py /tmp/v$i.json /tmp/v$i/ > /dev/null
  python - <<PY
import json
d=json.load(open("/tmp/v$i.json"))
for k,v in d.items():
    if "stream" in k: print("$v", k[30:70], {a.replace("_sum_per_launch",""):"%.3g"%b for a,b in v.items() if "TCC" in a})
PY
done
cd "$ROOT"
python - <<PY
import os, sys, subprocess, time
sys.path.insert(0, "$ROOT")
import bench
w = bench.Workload("C2", 100_000, 10_000, 1, 0, "db")
d = "/tmp/e2e"; os.makedirs(d, exist_ok=True)
q = w.write_fasta(d)
subprocess.run(["$ROOT/oracle/_ref/diamond", "makedb", "--in", d + "/db.faa", "-d", d + "/db", "-p", "16"], check=True, capture_output=True)
def run(flags, env):
    t0 = time.perf_counter()
    p = subprocess.run(["$ROOT/diamond_amd/diamond-hip", "blastp", "--fast", "-q", q, "-d", d + "/db", "-o", d + "/o.tsv"] + flags, capture_output=True, text=True, env=dict(os.environ, **env))
    return time.perf_counter() - t0, p.stderr
for flags in (["--masking", "0", "--motif-masking", "0", "--algo", "0"], ["--algo", "0"], []):
    res = [run(flags, {"DMND_CLI_TIMELINE": "1", "DMND_TRACE": "1"}) for _ in range(5)]
    print("== flags", flags, "wall", [round(r[0], 3) for r in res])
    print("\n".join(l for l in res[-1][1].splitlines() if l.startswith(("timeline", "dmnd_init", "Total"))))
PY
