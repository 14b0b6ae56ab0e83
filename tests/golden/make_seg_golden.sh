#!/bin/bash
# Golden of tests/test_seg.py: the reference's own SEG (oracle/_ref/seg_ref = src/lib/blast/blast_seg.cpp compiled in place behind
# oracle/seg_ref_main.cpp) on the reference's ctest fixture and on tests/golden/seg_cases.faa.gz (synthetic low-complexity cases:
# homopolymers, two-letter and biased regions, tandem repeats, non-standard letters, regions longer than the trim limit, sequences
# around the window length). One line per sequence: id, then begin-end (0-based, inclusive) of every masked segment.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
seg="$here/../../oracle/_ref/seg_ref"
{ "$seg" < "$here/ref_ctest/data.faa"; zcat "$here/seg_cases.faa.gz" | "$seg"; } | gzip -9nc > "$here/seg_golden.tsv.gz"
