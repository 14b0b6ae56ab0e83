#!/bin/bash
# Round 6: C5 with one / two seed contexts (batch-level seed tasks), and a 2-rank run of bench.py on ONE GPU (gloo exchange:
# DMND_BENCH_SHARE_GPU=1) to see the rccl block and the N > 1 parity leg work end to end
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r06i"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
show() {
python - "$1" "$2" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("{") and '"metric"' in line:
        d = json.loads(line)
        print(sys.argv[2], {k: d.get(k) for k in ("n_gpus", "ms_per_step", "value", "host_cpu_ms_per_step", "parity_checked")}, d.get("rccl"), d.get("scaling_model", {}).get("best"))
PY
}
for sc in 2 1 3; do
  timeout 900 python "$ROOT/bench.py" --config C5 --seed-contexts $sc --steps 12 --warmup 4 --no-cpu-baseline --no-e2e --no-masked-step > "$OUT/c5_sc$sc.json" 2> "$OUT/c5_sc$sc.err"; show "$OUT/c5_sc$sc.json" "C5 seed-contexts=$sc"
done
cd "$ROOT"
DMND_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --no-e2e --no-masked-step > "$OUT/c2_n2_shared.json" 2> "$OUT/c2_n2_shared.err"
show "$OUT/c2_n2_shared.json" "C2 N=2 on one GPU (gloo)"
tail -3 "$OUT/c2_n2_shared.err"
