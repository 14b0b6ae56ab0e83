#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r05d"; mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests/test_gpu_xdrop.py tests/test_gpu_skew.py -m gpu -q -x 2>&1 | cut -c1-1500 > "$OUT/pytest.txt"; grep -n "Error\|assert\|passed\|failed" "$OUT/pytest.txt" | head -40
