#!/bin/bash
# the GPU tests named in tools/quick_tests.txt (one pytest argument per line)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
timeout 1500 python -m pytest $(cat tools/quick_tests.txt) -m gpu -x -q 2>&1 | tail -8
