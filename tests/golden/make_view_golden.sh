#!/bin/bash
# Goldens of tests/test_view.py: a small blastx archive written by the reference (reads = back-translations of the first 40 proteins of
# the ctest fixture, diamond_amd.synth.back_translate seed 5; database = the fixture) and what the reference's own `view` prints from it
# and from the blastp archive tests/golden/daa_k4.daa.gz.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
root="$here/../.."
ref="$root/oracle/_ref/diamond"
tmp="$(mktemp -d)"
cd "$root"
python - "$tmp" <<'PY'
import sys
import numpy as np
from diamond_amd import synth
AA = "ARNDCQEGHILKMFPSTWYVBJZX*_"
recs = open("tests/golden/ref_ctest/data.faa").read().split(">")[1:41]
seqs = [np.array([AA.index(c) if c in AA[:20] else 0 for c in "".join(r.strip().split("\n")[1:]).upper()], np.int8) for r in recs]
off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])])
dna, doff = synth.back_translate(np.concatenate(seqs), off, seed=5)
synth.write_dna_fasta(sys.argv[1] + "/reads.fna", "read", dna, doff)
PY
"$ref" blastx -q "$tmp/reads.fna" -d "$here/ref_ctest/data.faa" -p 1 -k 3 -f 100 -o "$tmp/bx" --quiet
gzip -9nc "$tmp/bx.daa" > "$here/daa_blastx.daa.gz"
zcat "$here/daa_k4.daa.gz" > "$tmp/k4.daa"
{
  for f in "-f 6" "-f 0" "-f 103" "-f 6 qseqid sseqid qstart qend qframe qstrand btop cigar qcovhsp scovhsp positive gaps qseq_translated" "-f 6 -k 2" "-f 6 --top 3" "-f 6 --forwardonly"; do
    echo "#### bx $f"
    "$ref" view --daa "$tmp/bx.daa" $f --quiet 2>/dev/null
  done
  echo "#### bx -f 5"
  "$ref" view --daa "$tmp/bx.daa" -f 5 --quiet 2>/dev/null | grep -v "<BlastOutput_version>"; echo
  echo "#### bx -f 101"
  "$ref" view --daa "$tmp/bx.daa" -f 101 --quiet 2>/dev/null | grep -v "^@PG"
  for f in "-f 6 qseqid sseqid qlen slen score nident positive gapopen gaps ppos qcovhsp scovhsp qnum snum evalue bitscore" "-f 6 -k 2" "-f 6 --top 3"; do
    echo "#### k4 $f"
    "$ref" view --daa "$tmp/k4.daa" $f --quiet 2>/dev/null
  done
} | gzip -9nc > "$here/view_golden.txt.gz"
rm -r "$tmp"
