#!/bin/bash
# round 4: composition-based matrix adjustment on the GPU box -- the swipe parity tests (per-item matrices), the CLI A/B tests of
# --comp-based-stats 2..5, the bridge inside the genuine reference
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
timeout 1200 python -m pytest tests/test_gpu_swipe.py -m gpu -x -q -k "matrices" 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_gpu_cli.py -m gpu -x -q -k "comp_based or comp-based" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_reference_bridge.py -m gpu -x -q -k "cbs" 2>&1 | tail -5
