#!/bin/bash
# scatter + join for the short-seed stream (DMND_SEED_SJ=1): parity on the taps, timing on the C2 blocks, full-size C3 parity
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r05i"; mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python -m pytest tests/test_gpu_seed.py -m gpu -q -x -k "scatter_join or two_lane" 2>&1 | cut -c1-1500 | tail -25 > "$OUT/pytest.txt"; tail -6 "$OUT/pytest.txt"
cd /tmp && export TMPDIR=/tmp
sm() { env MODES_EXTEND=0 "$@" timeout 300 python "$ROOT/tools/seed_modes.py" sensitive 2 2>&1 | grep -v "^dmnd_seed_search [ 0-9.]* ms" | grep "sensitive\|spilled\|rror" | tail -3 | sed "s/^/$* : /" | tee -a "$OUT/sj.txt"; }
sm DMND_SEED_SJ=0
sm DMND_SEED_SJ=1
sm DMND_SEED_SJ=1 DMND_SEED_SLOTS_X8=16
sm DMND_SEED_SJ=1 DMND_SEED_SLOTS_X8=16 DMND_SEED_BM1_KB=3072 DMND_SEED_BM1_K=3
sm DMND_SEED_SJ=1 DMND_SEED_SLOTS_X8=16 DMND_SEED_BM1_KB=2048 DMND_SEED_BM1_K=2
cd "$ROOT"
DMND_SEED_SJ=1 DMND_SEED_SLOTS_X8=16 timeout 900 python -m pytest tests/test_gpu_fullscale.py -m gpu -q -x -k "c3_sensitive and 0" 2>&1 | tail -3 | tee -a "$OUT/sj.txt"
