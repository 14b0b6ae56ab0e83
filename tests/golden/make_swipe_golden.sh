#!/bin/bash
# Mints the kernel-level known answers under tests/golden/ from the GENUINE reference
# (oracle/_ref/diamond_tap = /root/reference compiled in place + the --wrap tap of oracle/ref_tap.cpp).
# Run in the build container only (needs /root/reference); the .tap fixtures are committed.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
REFTEST=/root/reference/src/test
TAP="$ROOT/oracle/_ref/diamond_tap"
TMP="$(mktemp -d)"
make -C "$ROOT/oracle" ref >/dev/null

# 1. default sensitivity on the reference's own 389-domain SCOP fixture (ctest diamond-test-blastp-default,
#    CMakeLists.txt:553): first 120 queries, round 1 (score only) + round 2 (traceback) calls
DIAMOND_TAP_FILE="$HERE/swipe_default.tap" DIAMOND_TAP_MAX_CALLS=240 \
  "$TAP" blastp -q "$REFTEST/data.faa" -d "$REFTEST/data.faa" -o "$TMP/default.out" -p1 2>/dev/null
diff -q "$TMP/default.out" "$REFTEST/diamond-test-blastp-default.out"   # the tap must not change results

# 2. --fast (BASELINE configs C1/C2) on the same fixture
DIAMOND_TAP_FILE="$HERE/swipe_fast.tap" DIAMOND_TAP_MAX_CALLS=200 \
  "$TAP" blastp --fast -q "$REFTEST/data.faa" -d "$REFTEST/data.faa" -o "$TMP/fast.out" -p1 2>/dev/null

# 3. long synthetic proteins: DP size > max_swipe_dp (1e6 cells, basic/config.cpp:595) forces the
#    statistics-without-traceback path (ForwardCell + recompute_reversed with BackwardCell)
DMND_ROOT="$ROOT" python3 - "$TMP" <<'PY'
import os, sys
sys.path.insert(0, os.environ["DMND_ROOT"])
from diamond_amd import synth
db, do, q, qo = synth.generate(3, members=2, queries=3, len_mean=9000, len_sd=500, len_min=8000, len_max=10000,
                               seed=7, decoy_frac=0.0)
synth.write_fasta(sys.argv[1] + "/long_db.faa", "t", db, do)
synth.write_fasta(sys.argv[1] + "/long_q.faa", "q", q, qo)
PY
DIAMOND_TAP_FILE="$HERE/swipe_long.tap" \
  "$TAP" blastp -q "$TMP/long_q.faa" -d "$TMP/long_db.faa" -o "$TMP/long.out" -p1 2>/dev/null

# 4. blastx (6 query contexts, frames) on the reference's galaxy fixture (ctest galaxy_7)
DIAMOND_TAP_FILE="$HERE/swipe_blastx.tap" \
  "$TAP" blastx -q "$REFTEST/galaxy/nucleotide.fasta" -d "$REFTEST/galaxy/db.dmnd" -o "$TMP/bx.out" -p1 2>/dev/null || true
# 5. seed stage + extension stage known answers, tapped at Extension::extend (align/extend.cpp:346): per query the
#    stage-2 seed hits it receives and the Match list it returns; header = both sequence blocks + seed configuration.
#    Masking is host pre-processing outside the path: run with --masking 0 --motif-masking 0.
DIAMOND_TAP_EXT="$HERE/ext_fast.tap" \
  "$TAP" blastp --fast --masking 0 --motif-masking 0 --algo 0 -q "$REFTEST/data.faa" -d "$REFTEST/data.faa" -o "$TMP/e1.out" -p1 2>/dev/null
# six shapes of weight 10 (still no ungapped filter): exercises the cross-shape left-most rule and mask times
DIAMOND_TAP_EXT="$HERE/ext_6x10.tap" \
  "$TAP" blastp --shapes-6x10 --masking 0 --motif-masking 0 --algo 0 -q "$REFTEST/data.faa" -d "$REFTEST/data.faa" -o "$TMP/e6.out" -p4 2>/dev/null
DMND_ROOT="$ROOT" python3 - "$TMP" <<'PY'
import os, sys
sys.path.insert(0, os.environ["DMND_ROOT"])
from diamond_amd import synth
db, do, q, qo = synth.generate(250, members=10, queries=300, seed=1)
synth.write_fasta(sys.argv[1] + "/s_db.faa", "t", db, do)
synth.write_fasta(sys.argv[1] + "/s_q.faa", "q", q, qo)
PY
# both seams in one run: the same queries' seed hits, DpTargets (= band geometry from chaining) and Match lists;
# the reference's TSV output is kept as the end-to-end golden
DIAMOND_TAP_EXT="$HERE/ext_fast_synth.tap" DIAMOND_TAP_FILE="$HERE/swipe_fast_synth.tap" \
  "$TAP" blastp --fast --masking 0 --motif-masking 0 --algo 0 -q "$TMP/s_q.faa" -d "$TMP/s_db.faa" -o "$HERE/fast_synth.tsv" -p4 2>/dev/null
# 6. ranking chunks: a few large, well-conserved families so that every query has far more than 128 seed-hit targets
#    (ranking_chunk_size, extend.cpp:79-92): exercises the chunked target ranking / early termination loop
DMND_ROOT="$ROOT" python3 - "$TMP" <<'PY'
import os, sys
sys.path.insert(0, os.environ["DMND_ROOT"])
from diamond_amd import synth
db, do, q, qo = synth.generate(3, members=400, queries=24, seed=9, sub=(0.05, 0.5), qsub=(0.05, 0.3), decoy_frac=0.0)
synth.write_fasta(sys.argv[1] + "/r_db.faa", "t", db, do)
synth.write_fasta(sys.argv[1] + "/r_q.faa", "q", q, qo)
PY
DIAMOND_TAP_EXT="$HERE/ext_rank.tap" \
  "$TAP" blastp --fast --masking 0 --motif-masking 0 --algo 0 -q "$TMP/r_q.faa" -d "$TMP/r_db.faa" -o "$HERE/rank.tsv" -p1 2>/dev/null
# 7. default sensitivity (two weight-10 shapes + the stage-2 ungapped e-value filter): the reference's
#    own fixture, and the synthetic block (whose conserved families push window scores past 255 -> int8 saturation rule)
DIAMOND_TAP_EXT="$HERE/ext_default.tap" \
  "$TAP" blastp --masking 0 --motif-masking 0 --algo 0 -q "$REFTEST/data.faa" -d "$REFTEST/data.faa" -o "$HERE/default.tsv" -p1 2>/dev/null
DIAMOND_TAP_EXT="$HERE/ext_default_synth.tap" \
  "$TAP" blastp --masking 0 --motif-masking 0 --algo 0 -q "$TMP/s_q.faa" -d "$TMP/s_db.faa" -o "$HERE/default_synth.tsv" -p4 2>/dev/null
# 8. --sensitive (16 shapes of weight 8 + the gapped filter): third seam = Extension::gapped_filter
#    (per query: Hauser bias, the seed hits of every target of the ranking chunk, both cutoffs, the surviving targets)
DIAMOND_TAP_EXT="$HERE/ext_sensitive.tap" DIAMOND_TAP_GF="$HERE/gf_sensitive.tap" \
  "$TAP" blastp --sensitive --masking 0 --motif-masking 0 --algo 0 -q "$REFTEST/data.faa" -d "$REFTEST/data.faa" -o "$HERE/sensitive.tsv" -p1 2>/dev/null
# 9. blastx: synthetic DNA reads (back-translated protein queries with random codons, flanks and strand) against a
#    protein DB; the tap header holds the TRANSLATED query block (6 frames per read, short ORFs masked), so the golden also
#    pins dmnd_translate. The reads are kept as blastx_reads.fna (source lengths for the DNA coordinates of the output).
DMND_ROOT="$ROOT" python3 - "$TMP" "$HERE" <<'PY'
import os, sys
sys.path.insert(0, os.environ["DMND_ROOT"])
from diamond_amd import synth
db, do, q, qo = synth.generate(150, members=10, queries=120, seed=3)
dna, off = synth.back_translate(q, qo, seed=4)
synth.write_fasta(sys.argv[1] + "/x_db.faa", "t", db, do)
synth.write_dna_fasta(sys.argv[2] + "/blastx_reads.fna", "r", dna, off)
PY
DIAMOND_TAP_EXT="$HERE/ext_blastx.tap" \
  "$TAP" blastx --masking 0 --motif-masking 0 --algo 0 -q "$HERE/blastx_reads.fna" -d "$TMP/x_db.faa" -o "$HERE/blastx.tsv" -p1 2>/dev/null
# 10. the long proteins of step 3 end to end (round 2 takes the statistics-without-traceback path: DP size > 1e6 cells)
DIAMOND_TAP_EXT="$HERE/ext_long.tap" \
  "$TAP" blastp --masking 0 --motif-masking 0 --algo 0 -q "$TMP/long_q.faa" -d "$TMP/long_db.faa" -o "$HERE/long.tsv" -p1 2>/dev/null
# 11. duplicate query seeds with ambiguity letters: 16-letter windows that differ only by A vs B / J / Z at a
#     care position of the --fast shape (same reduced seed, but seed_is_complex rejects the window with the ambiguity letter),
#     as query pairs in both file orders; targets hold the A window. Queries / database are kept as bjz_q.faa / bjz_db.faa.
python3 - "$HERE" <<'PY'
import sys
import numpy as np
rng = np.random.default_rng(42)
AA = "ARNDCQEGHILKMFPSTWYV"
rnd = lambda n: "".join(AA[i] for i in rng.integers(0, 20, n))
care = [0, 1, 3, 4, 5, 7, 9, 10, 12, 13, 14, 15]
qs, ts, n = [], [], 0
for amb in "BJZ":
    for typ in (1, 2):
        for rep in range(20):
            cp = care[rng.integers(0, len(care))]
            w = list(rnd(16)); w[cp] = "A"; W = "".join(w)
            w2 = list(W); w2[cp] = amb; W2 = "".join(w2)
            a, b = rnd(40) + W + rnd(40), rnd(40) + W2 + rnd(40)
            for s in ((a, b) if typ == 1 else (b, a)):
                qs.append((">q%d_%s_t%d" % (n, amb, typ), s)); n += 1
            ts.append((">t%d" % len(ts), rnd(60) + W + rnd(60)))
for i in range(300):
    ts.append((">r%d" % i, rnd(int(rng.integers(100, 400)))))
open(sys.argv[1] + "/bjz_q.faa", "w").write("".join("%s\n%s\n" % x for x in qs))
open(sys.argv[1] + "/bjz_db.faa", "w").write("".join("%s\n%s\n" % x for x in ts))
PY
DIAMOND_TAP_EXT="$HERE/ext_bjz.tap" \
  "$TAP" blastp --fast --masking 0 --motif-masking 0 --algo 0 -q "$HERE/bjz_q.faa" -d "$HERE/bjz_db.faa" -o "$TMP/bjz.out" -p1 2>/dev/null
ls -la "$HERE"/*.tap
rm -rf "$TMP"
