#!/bin/bash
# Goldens of the output formats (tests/test_format.py): the reference binary on its own ctest fixture, all against all, -k 4.
# usage: tests/golden/make_format_golden.sh   (needs oracle/_ref/diamond; run from anywhere)
set -e
here="$(cd "$(dirname "$0")" && pwd)"
ref="$here/../../oracle/_ref/diamond"
cd "$here/ref_ctest"
fields="qseqid qlen sseqid sallseqid slen qstart qend sstart send qseq sseq evalue bitscore score length pident nident mismatch positive gapopen gaps ppos qframe btop stitle salltitles qcovhsp qtitle full_sseq qnum snum scovhsp full_qseq qseq_gapped sseq_gapped qstrand cigar"
tmp="$(mktemp -d)"
"$ref" blastp -q data.faa -d data.faa -p 4 -k 4 -f 6 $fields -o "$tmp/fields.tsv" --quiet
"$ref" blastp -q data.faa -d data.faa -p 4 -k 4 -f 0 -o "$tmp/pairwise.out" --quiet
"$ref" blastp -q data.faa -d data.faa -p 4 -k 4 -f paf -o "$tmp/paf.out" --quiet
"$ref" blastp -q data.faa -d data.faa -p 4 -k 4 -f sam -o "$tmp/sam.out" --quiet
"$ref" blastp -q data.faa -d data.faa -p 4 -k 4 -f 5 -o "$tmp/xml.out" --quiet
gzip -9nc "$tmp/fields.tsv" > "$here/fields_k4.tsv.gz"
gzip -9nc "$tmp/pairwise.out" > "$here/pairwise_k4.out.gz"
gzip -9nc "$tmp/paf.out" > "$here/paf_k4.out.gz"
grep -v '^@' "$tmp/sam.out" | gzip -9nc > "$here/sam_k4.body.gz"
gzip -9nc "$tmp/xml.out" > "$here/xml_k4.out.gz"
"$ref" blastp -q data.faa -d data.faa -p 1 -k 4 -f 100 -o "$tmp/aln" --quiet
gzip -9nc "$tmp/aln.daa" > "$here/daa_k4.daa.gz"
rm -r "$tmp"
