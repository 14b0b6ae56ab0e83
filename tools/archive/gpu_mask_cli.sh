#!/bin/bash
# round 4: masking phases of `diamond-hip blastp --algo 0` on the C2 files: host laps (DMND_TRACE) and the kernels' times (rocprofv3 --stats)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; OUT="$ROOT/gpurun_out/e2e"; mkdir -p "$OUT"; T=/tmp/e2e; mkdir -p $T
python - <<PY
import sys; sys.path.insert(0, "$ROOT")
import bench
w = bench.Workload("C2", 100000, 10000, 1, 0, "db")
w.write_fasta("$T")
PY
"$ROOT/diamond_amd/diamond-hip" makedb --in $T/db.faa -d $T/db > /dev/null 2>&1
Q=$(ls $T/*.faa | grep -v db.faa | head -1)
DMND_TRACE=1 DMND_CLI_TIMELINE=1 timeout 120 "$ROOT/diamond_amd/diamond-hip" blastp --algo 0 -q $Q -d $T/db -o $T/o.tsv 2>&1 | grep -E "dmnd_mask|dmnd_soft|timeline.*mask"
cd /tmp && export TMPDIR=/tmp
DMND_CLI_CLEAN_EXIT=1 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $T/st -o s -- "$ROOT/diamond_amd/diamond-hip" blastp --algo 0 -q $Q -d $T/db -o $T/o.tsv > /dev/null 2>&1
f=$(find $T/st -name "*kernel_stats.csv" | head -1); grep -iE "motif|tantan|copyBuffer|fillBuffer" $f | cut -d, -f1-4 | cut -c1-120
