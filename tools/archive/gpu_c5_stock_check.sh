#!/bin/bash
# round 4: the stock command line on the C5 files after the parallel seed count of dmnd_auto_query_indexed: same output as the reference, wall times
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; OUT="$ROOT/gpurun_out/e2e"; mkdir -p "$OUT"; T=/tmp/e2e5; mkdir -p $T
timeout 200 python -m pytest tests/test_gpu_mask.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python - <<PY
import sys; sys.path.insert(0, "$ROOT")
import bench
w = bench.Workload("C5", 500000, 100000, 1, 0, "db")
w.write_fasta("$T")
open("$T/b.txt", "w").write("%.12f" % ((w.block_letters + 0.5) / 1e9))
PY
REF="$ROOT/oracle/_ref/diamond"
timeout 200 $REF makedb --in $T/db.faa -d $T/db -p 16 > /dev/null 2>&1
Q=$(ls $T/*.faa | grep -v db.faa | head -1)
s=$(date +%s.%N); timeout 300 $REF blastp -q $Q -d $T/db -o $T/ref.tsv -b $(cat $T/b.txt) -p 16 > $T/ref.log 2>&1; e=$(date +%s.%N)
echo "reference stock wall $(python -c "print(round($e-$s,3))")"
for rep in 1 2 3; do
  sleep 1
  s=$(date +%s.%N); DMND_CLI_TIMELINE=1 timeout 120 "$ROOT/diamond_amd/diamond-hip" blastp -q $Q -d $T/db -o $T/o.tsv -b $(cat $T/b.txt) > $T/log.txt 2>&1; e=$(date +%s.%N)
  echo "ours stock wall $(python -c "print(round($e-$s,3))") $(grep -E 'Total time|Algorithm' $T/log.txt | tr '\n' ' ')"
done
md5sum $T/ref.tsv $T/o.tsv | awk '{print $1}' | uniq | wc -l
grep -E "timeline" $T/log.txt | head -8 > "$OUT/timeline_C5_stock_after_fix.txt"; head -6 "$OUT/timeline_C5_stock_after_fix.txt"
