#!/bin/bash
# Mints the known answers of composition-based matrix adjustment (--comp-based-stats 2..5) under tests/golden/ from the
# GENUINE reference (oracle/_ref/diamond_tap, fifth seam of oracle/ref_tap.cpp: Stats::adjust_matrix and
# Stats::TargetMatrix::TargetMatrix, src/stats/cbs.cpp:94-173), plus DP::BandedSwipe::swipe batches whose DpTargets carry their
# adjusted matrices ('SWP2' records). Run in the build container only (needs /root/reference); the fixtures are committed.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
REFTEST=/root/reference/src/test
TAP="$ROOT/oracle/_ref/diamond_tap"
TMP="$(mktemp -d)"
make -C "$ROOT/oracle" ref >/dev/null
run() { "$TAP" blastp -q "$REFTEST/data.faa" -d "$REFTEST/data.faa" -p1 "$@" 2>/dev/null; }

# 1. rule decisions + adjusted matrices on the reference's own ctest command lines (CMakeLists.txt:563-565); the outputs must be
#    the reference's goldens (the tap does not change results)
for m in 3 4; do
  DIAMOND_TAP_CBS="$HERE/cbs_mode$m.tap" DIAMOND_TAP_CBS_MAX=900 run --more-sensitive -c1 --comp-based-stats $m -o "$TMP/m$m.out"
  diff -q "$TMP/m$m.out" "$REFTEST/diamond-test-blastp-comp-based-stats-$m.out"
done
# 2. mode 5: the lambda-rescaling rule (CompositionBasedStats, comp_based_stats.cpp:402-460) beside the full adjustment
DIAMOND_TAP_CBS="$HERE/cbs_mode5.tap" DIAMOND_TAP_CBS_MAX=900 run --more-sensitive -c1 --comp-based-stats 5 -o "$TMP/m5.out"
# 3. other matrices: their own joint probabilities, background frequencies and ideal lambda
DIAMOND_TAP_CBS="$HERE/cbs_blosum45.tap" DIAMOND_TAP_CBS_MAX=400 run --comp-based-stats 5 --matrix BLOSUM45 -o "$TMP/b45.out"
DIAMOND_TAP_CBS="$HERE/cbs_pam70.tap" DIAMOND_TAP_CBS_MAX=400 run --comp-based-stats 4 --matrix PAM70 -o "$TMP/p70.out"
# 4. sweeps with per-target matrices: mode 3 mixes adjusted targets (no Hauser bias in the sweep) with unadjusted ones (bias),
#    mode 4 adjusts every target
DIAMOND_TAP_MATRICES=1 DIAMOND_TAP_FILE="$HERE/swipe_cbs3.tap" DIAMOND_TAP_MAX_CALLS=160 run --comp-based-stats 3 -o "$TMP/s3.out"
DIAMOND_TAP_MATRICES=1 DIAMOND_TAP_FILE="$HERE/swipe_cbs4.tap" DIAMOND_TAP_MAX_CALLS=120 run --comp-based-stats 4 -o "$TMP/s4.out"
# 5. the ctest goldens themselves, for the CLI tests on the GPU box
for m in 2 3 4; do cp "$REFTEST/diamond-test-blastp-comp-based-stats-$m.out" "$HERE/ref_ctest/"; done
ls -la "$HERE"/cbs_*.tap "$HERE"/swipe_cbs*.tap
rm -rf "$TMP"
