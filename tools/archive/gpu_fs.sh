#!/bin/bash
# round 4: frameshift alignment on the GPU box -- kernel parity, then the command-line A/B against the reference binary
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
timeout 600 python -m pytest tests/test_gpu_frameshift.py -m gpu -x -q 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_cli.py -m gpu -x -q -k "frameshift" 2>&1 | tail -25
