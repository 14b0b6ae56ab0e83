#!/bin/bash
# the GPU tests named in tools/quick_tests.txt (the arguments of one pytest command line)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
eval "timeout 1500 python -m pytest $(cat tools/quick_tests.txt) -m gpu -x -q" 2>&1 | tail -8
