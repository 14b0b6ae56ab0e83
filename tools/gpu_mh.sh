#!/bin/bash
# --max-hsps debugging: reference and diamond-hip on the multi-domain test data, outputs kept under gpurun_out/mh
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/mh"; mkdir -p "$OUT"; cd "$ROOT"
python - <<PY
import sys, importlib.util, pathlib
sys.path.insert(0, "$ROOT"); sys.path.insert(0, "$ROOT/tests")
spec = importlib.util.spec_from_file_location("tgc", "$ROOT/tests/test_gpu_cli.py")
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
m._write_multi_hsp_files(pathlib.Path("$OUT"))
PY
i=0
while read -r args; do
  i=$((i+1))
  oracle/_ref/diamond blastp -q $OUT/q.faa -d $OUT/db.faa -p 4 $args -o $OUT/ref_$i.out 2>/dev/null
  diamond_amd/diamond-hip blastp -q $OUT/q.faa -d $OUT/db.faa -p 4 $args -o $OUT/hip_$i.out 2> $OUT/hip_$i.err || tail -3 $OUT/hip_$i.err
  echo "== $args: ref $(wc -l < $OUT/ref_$i.out) hip $(wc -l < $OUT/hip_$i.out) differing lines $(diff $OUT/ref_$i.out $OUT/hip_$i.out | grep -c '^[<>]')"
  diff $OUT/ref_$i.out $OUT/hip_$i.out | head -${DIFF_LINES:-8}
done < tools/mh_cases.txt
