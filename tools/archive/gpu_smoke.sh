#!/bin/bash
# the driver's round-end entry points on the GPU box: __graft_entry__.smoke(), then bench.py with no flags
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 300 python bench.py --no-e2e --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300
