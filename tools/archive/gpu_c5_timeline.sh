#!/bin/bash
# round 4: timeline of the stock command line on the C5 files (8 reference blocks): where do 2.5 s outside the logged stages go?
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; OUT="$ROOT/gpurun_out/e2e"; mkdir -p "$OUT"; T=/tmp/e2e5; mkdir -p $T
python - <<PY
import sys; sys.path.insert(0, "$ROOT")
import bench
w = bench.Workload("C5", 500000, 100000, 1, 0, "db")
print(w.write_fasta("$T"), w.n_blocks_total, w.block_letters)
open("$T/b.txt", "w").write("%.12f" % ((w.block_letters + 0.5) / 1e9))
PY
"$ROOT/diamond_amd/diamond-hip" makedb --in $T/db.faa -d $T/db > /dev/null 2>&1
Q=$(ls $T/*.faa | grep -v db.faa | head -1)
for rep in 1 2; do
  sleep 1
  DMND_CLI_TIMELINE=1 "$ROOT/diamond_amd/diamond-hip" blastp -q $Q -d $T/db -o $T/o.tsv -b $(cat $T/b.txt) > $T/log.txt 2>&1
done
grep -E "timeline|Total" $T/log.txt | tee "$OUT/timeline_C5_stock.txt" | head -150
