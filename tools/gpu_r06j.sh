#!/bin/bash
# Round 6: start-up floor of a HIP process (init_probe) and the first dmnd_extend of a diamond-hip run, phase by phase
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; OUT="$ROOT/gpurun_out/r06j"; mkdir -p "$OUT"; T=/tmp/e2e; mkdir -p $T
for i in 1 2 3; do tools/probes/bin/init_probe | tr '\n' ';'; echo; done | tee "$OUT/init_probe.txt"
python - <<PY
import sys; sys.path.insert(0, "$ROOT")
import bench
w = bench.Workload("C2", 100000, 10000, 1, 0, "db")
print(w.write_fasta("$T"))
PY
"$ROOT/diamond_amd/diamond-hip" makedb --in $T/db.faa -d $T/db > /dev/null 2>&1
Q=$(ls $T/*.faa | grep -v db.faa | head -1)
for rep in 1 2 3; do
  sleep 1
  s=$(date +%s.%N)
  DMND_TRACE=1 DMND_CLI_TIMELINE=1 "$ROOT/diamond_amd/diamond-hip" blastp -q $Q -d $T/db -o $T/o.tsv > $T/log.txt 2>&1
  e=$(date +%s.%N)
  echo "=== rep $rep wall $(python -c "print(round($e-$s,4))")" | tee -a "$OUT/timeline.txt"
  if [ $rep = 3 ]; then grep -E "timeline|Total|dmnd_init|dmnd_extend" $T/log.txt | tee -a "$OUT/timeline.txt"; fi
done
