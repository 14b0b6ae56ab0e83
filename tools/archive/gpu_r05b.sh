#!/bin/bash
# Round 5, second GPU call: the whole GPU suite (all failures listed), the masking probe under a kernel trace, the C2 residency-cap x
# seed-context sweep, the C3 knob sweep on the seed stage alone, C5 with and without the by-class stream for long seeds.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r05b"; mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 > "$OUT/pytest.txt"; tail -4 "$OUT/pytest.txt"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/mask_stats" -o s -- python "$ROOT/tools/mask_probe.py" 3 > "$OUT/mask_probe.txt" 2>&1
grep MASK_PROBE "$OUT/mask_probe.txt"
find "$OUT/mask_stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_mask.csv"; rm -rf "$OUT/mask_stats"
grep -iE "motif|tantan" "$OUT/kernel_stats_mask.csv" | cut -d, -f1-4 | cut -c1-160
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], 'ms/step %.3f median %.3f seed_ms %s stream in pipeline %.3f hits/parity %s' % (d['ms_per_step'], d.get('ms_per_step_median') or 0, [round(x,3) for x in d['alone']['seed_kernel_ms']], d['roofline']['kernel_ms'], d.get('parity_checked')))" "$1"; }
for sc in 1 2; do for wg in 0 2 3 4; do
  DMND_SEED_STREAM_WGS=$wg timeout 200 python "$ROOT/bench.py" --steps 40 --warmup 8 --same-block --seed-contexts $sc --no-cpu-baseline --no-masked-step 2>/dev/null | line "C2 seed-contexts=$sc stream-wgs=$wg" | tee -a "$OUT/c2_overlap.txt"
done; done
DMND_SEED_STREAM_WGS=3 timeout 200 python "$ROOT/bench.py" --steps 40 --warmup 8 --no-cpu-baseline --no-masked-step 2>/dev/null | line "C2 alternating stream-wgs=3" | tee -a "$OUT/c2_overlap.txt"
c3() { env MODES_EXTEND=0 DMND_TRACE= "$@" timeout 200 python "$ROOT/tools/seed_modes.py" sensitive 2 2>/dev/null | grep -v "^dmnd" | tail -1 | sed "s/^/$* : /" | tee -a "$OUT/c3_knobs.txt"; }
c3 DMND_SEED_TFOLD=0
c3 DMND_SEED_TFOLD=1
c3 DMND_SEED_TFOLD=1 DMND_SEED_STREAM_NT=1
c3 DMND_SEED_TFOLD=1 DMND_SEED_SLOTS_X8=16
c3 DMND_SEED_TFOLD=1 DMND_SEED_BITMAP1_LOG2=25 DMND_SEED_BM1_K=3
c3 DMND_SEED_TFOLD=1 DMND_SEED_BITMAP1_LOG2=25 DMND_SEED_BM1_K=3 DMND_SEED_SLOTS_X8=16
c3 DMND_SEED_TFOLD=1 DMND_SEED_BITMAP1_LOG2=26 DMND_SEED_BM1_K=3
for cl in 0 1; do
  DMND_SEED_CLASSES_LONG=$cl timeout 400 python "$ROOT/bench.py" --config C5 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | line "C5 classes-long=$cl" | tee -a "$OUT/c5_classes.txt"
done
