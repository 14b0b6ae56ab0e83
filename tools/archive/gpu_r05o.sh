#!/bin/bash
# round 5: the seed stage without a host wait per timed phase (deferred event spans) and with the counts read together;
# DMND_SYNC_SPIN_US = bounded polling before the interrupt-driven wait. A/B on one box, interleaved.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r05o"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], 'ms/step %.3f median %.3f | seed call alone %.3f in pipeline p50 %.3f | ext p50 %.2f | host cpu %.1f' % (d['ms_per_step'], d.get('ms_per_step_median') or 0, d['alone']['seed_stage_call_ms'], d['latency_in_pipeline']['seed_stage_call_ms']['p50'], d['latency_in_pipeline']['extension_of_a_batch_ms']['p50'], d['host_cpu_ms_per_step']))" "$1"; }
for rep in 1 2 3; do
for v in "1 0" "0 0" "0 40" "0 150"; do
  set -- $v
  DMND_SEED_TIMER_WAITS=$1 DMND_SYNC_SPIN_US=$2 timeout 300 python "$ROOT/bench.py" --config C2 --steps 90 --warmup 15 --no-e2e --no-masked-step --no-cpu-baseline > "$OUT/bench_C2_w$1_s$2_$rep.json" 2>/dev/null; line "C2 waits $1 spin $2 rep $rep" < "$OUT/bench_C2_w$1_s$2_$rep.json"
done
done
