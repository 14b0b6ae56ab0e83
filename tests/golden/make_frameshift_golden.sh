#!/bin/bash
# Mints the known answers of the three-frame banded sweep (frameshift alignment, blastx -F) from the GENUINE reference
# (oracle/_ref/diamond_tap, sixth seam of oracle/ref_tap.cpp: banded_3frame_swipe, src/dp/dp.h:296) on synthetic reads with
# single-base insertions and deletions, and the reference's output of the same runs for the command-line tests.
# Run in the build container only (needs /root/reference); the fixtures are committed.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
TAP="$ROOT/oracle/_ref/diamond_tap"
TMP="$(mktemp -d)"
make -C "$ROOT/oracle" ref >/dev/null
DMND_ROOT="$ROOT" python3 - "$HERE" <<'PY'
import os, sys
sys.path.insert(0, os.environ["DMND_ROOT"])
import numpy as np
from diamond_amd import synth
db, doff, q, qoff = synth.generate(60, members=8, queries=90, seed=71)
synth.write_fasta(sys.argv[1] + "/fs_db.faa", "t", db, doff)
dna, off = synth.back_translate(q, qoff, seed=72)
synth.write_dna_fasta(sys.argv[1] + "/fs_reads.fna", "r", *synth.indel_reads(dna, off, seed=73))
PY
# -k 3: more targets than reported ones -> the score-only sweep (16 targets per vector on one band geometry) and its culling run
DIAMOND_TAP_3F="$HERE/f3_k3.tap" "$TAP" blastx -q "$HERE/fs_reads.fna" -d "$HERE/fs_db.faa" -F 15 -k 3 -o "$HERE/fs_k3.tsv" -p1 2>/dev/null
DIAMOND_TAP_3F="$HERE/f3_k1.tap" "$TAP" blastx -q "$HERE/fs_reads.fna" -d "$HERE/fs_db.faa" -F 15 -k 1 --sensitive -o "$TMP/k1.tsv" -p1 2>/dev/null
"$ROOT/oracle/_ref/diamond" blastx -q "$HERE/fs_reads.fna" -d "$HERE/fs_db.faa" -F 15 -o "$HERE/fs_f15.tsv" -p1 2>/dev/null
"$ROOT/oracle/_ref/diamond" blastx -q "$HERE/fs_reads.fna" -d "$HERE/fs_db.faa" -F 15 --range-culling --top 10 -o "$HERE/fs_f15_rc.tsv" -p1 2>/dev/null
ls -la "$HERE"/f3_k3.tap "$HERE"/fs_*
rm -rf "$TMP"
# the archive of the same run and what the reference's own `view` prints of it (tests/test_view.py::test_view_of_frameshift_alignments)
TMP2="$(mktemp -d)"
"$ROOT/oracle/_ref/diamond" blastx -q "$HERE/fs_reads.fna" -d "$HERE/fs_db.faa" -F 15 -f 100 -o "$TMP2/fs.daa" -p 2 2>/dev/null
"$ROOT/oracle/_ref/diamond" view -a "$TMP2/fs.daa" -f 6 qseqid sseqid pident length mismatch gapopen qstart qend sstart send evalue bitscore qframe nident positive gaps btop cigar qseq_gapped sseq_gapped sseq qcovhsp -o "$TMP2/v.tsv" 2>/dev/null
gzip -9 -c "$TMP2/fs.daa" > "$HERE/fs_f15.daa.gz"; gzip -9 -c "$TMP2/v.tsv" > "$HERE/fs_f15_view_fields.tsv.gz"; rm -rf "$TMP2"
# ... and in the pairwise, XML, PAF and SAM formats (tests/golden/fs_f15_view_formats.txt.gz: sections "#### -f N")
TMP3="$(mktemp -d)"; gzip -dc "$HERE/fs_f15.daa.gz" > "$TMP3/fs.daa"
for f in 0 5 paf 101; do echo "#### -f $f"; "$ROOT/oracle/_ref/diamond" view -a "$TMP3/fs.daa" -f $f 2>/dev/null; done | gzip -9 > "$HERE/fs_f15_view_formats.txt.gz"; rm -rf "$TMP3"
