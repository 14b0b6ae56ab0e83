#!/bin/bash
# round 4: where the wall time of a whole diamond-hip process goes (C2 files): DMND_CLI_TIMELINE + DMND_TRACE for the three command lines of the e2e leg
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; OUT="$ROOT/gpurun_out/e2e"; mkdir -p "$OUT"; T=/tmp/e2e; mkdir -p $T
python - <<PY
import sys; sys.path.insert(0, "$ROOT")
import bench
w = bench.Workload("C2", 100000, 10000, 1, 0, "db")
print(w.write_fasta("$T"))
PY
"$ROOT/diamond_amd/diamond-hip" makedb --in $T/db.faa -d $T/db > /dev/null 2>&1
Q=$(ls $T/*.faa | grep -v db.faa | head -1)
for flags in "" "--algo 0"; do
  for rep in 1 2 3; do
    sleep 1
    s=$(date +%s.%N)
    DMND_TRACE=1 DMND_CLI_TIMELINE=1 "$ROOT/diamond_amd/diamond-hip" blastp $flags -q $Q -d $T/db -o $T/o.tsv > $T/log.txt 2>&1
    e=$(date +%s.%N)
    echo "=== flags='$flags' rep $rep wall $(python -c "print(round($e-$s,4))")" | tee -a "$OUT/timeline.txt"
    if [ $rep = 3 ]; then grep -E "timeline|Total|dmnd_init|dmnd_extend|dmnd_mask" $T/log.txt | tee -a "$OUT/timeline.txt"; fi
  done
done
