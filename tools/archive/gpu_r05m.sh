#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
env MODES_EXTEND=0 DMND_SEED_SJ=1 DMND_SEED_SLOTS_X8=16 DMND_SEED_PHASES=1 timeout 300 python "$ROOT/tools/seed_modes.py" sensitive 1 2>&1 | grep "SEED_PHASES\|^sensitive" | cut -c1-300
