#!/bin/bash
# Round 5, third GPU call: the new tests (device join, RCCL merge, skewed workload, by-class long seeds, blocked frameshift, cbs passes),
# then C5 (by-class stream + device join, parity inside the run), C2 (masked step with the new motif kernels), C3, C2skew.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r05c"; mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests/test_gpu_join_device.py tests/test_gpu_skew.py tests/test_gpu_seed.py tests/test_gpu_db_shard.py tests/test_gpu_mask.py tests/test_gpu_cli.py -m gpu -q \
  -k "join or skew or seed or shard or mask or gpus or frameshift_blocked or comp_based_stats_matrix or cli_blocked" 2>&1 | tail -40 > "$OUT/pytest.txt"; tail -5 "$OUT/pytest.txt"
cd /tmp && export TMPDIR=/tmp
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], 'ms/step %.3f median %.3f seed_ms %s stream in pipeline %.3f parity %s host_cpu %.1f' % (d['ms_per_step'], d.get('ms_per_step_median') or 0, [round(x,3) for x in d['alone']['seed_kernel_ms']], d['roofline']['kernel_ms'], d.get('parity_checked'), d['host_cpu_ms_per_step']))
if 'masked_step' in d: print('   masked', round(d['masked_step']['ms_per_step'],2), d['masked_step']['parts_ms'], d['masked_step'].get('parity',{}).get('matches'))
if 'e2e' in d: print('   e2e', {k:(round(v['speedup'],2), round(v['speedup_min'],2), v['parity']) for k,v in d['e2e']['runs'].items()})" "$1"; }
timeout 600 python "$ROOT/bench.py" --config C5 --steps 6 --warmup 2 --no-e2e > "$OUT/bench_C5.json" 2> "$OUT/bench_C5.err"; line C5 < "$OUT/bench_C5.json"; tail -c 300 "$OUT/bench_C5.err"
timeout 300 python "$ROOT/bench.py" --config C5 --steps 6 --warmup 2 --no-cpu-baseline --host-join > "$OUT/bench_C5_host_join.json" 2>/dev/null; line "C5 host join" < "$OUT/bench_C5_host_join.json"
timeout 600 python "$ROOT/bench.py" --steps 30 --warmup 6 > "$OUT/bench_C2.json" 2> "$OUT/bench_C2.err"; line C2 < "$OUT/bench_C2.json"
timeout 600 python "$ROOT/bench.py" --config C3 --steps 6 --warmup 2 --no-e2e > "$OUT/bench_C3.json" 2> "$OUT/bench_C3.err"; line C3 < "$OUT/bench_C3.json"
timeout 500 python "$ROOT/bench.py" --config C2skew --steps 6 --warmup 2 --no-e2e > "$OUT/bench_C2skew.json" 2> "$OUT/bench_C2skew.err"; line C2skew < "$OUT/bench_C2skew.json"; tail -c 300 "$OUT/bench_C2skew.err"
