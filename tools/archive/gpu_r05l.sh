#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
timeout 300 python -m pytest tests/test_gpu_seed.py -m gpu -q -x -k "scatter_join" 2>&1 | tail -3
bash tools/gpu_r05j.sh
