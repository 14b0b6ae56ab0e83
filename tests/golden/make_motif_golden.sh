#!/bin/bash
# Mints the goldens of the motif soft masking from the GENUINE reference run with its DEFAULT flags (tantan + motif masking):
# synthetic sequences with motifs of the reference's table planted in them; stage-2 hits tapped at Extension::extend for the
# double-indexed and the query-indexed algorithm. Needs /root/reference (build container) and diamond_amd/motifs.bin.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"; ROOT="$(cd "$HERE/../.." && pwd)"; TAP="$ROOT/oracle/_ref/diamond_tap"; TMP="$(mktemp -d)"
python3 "$ROOT/tools/make_motif_table.py" >/dev/null
DMND_ROOT="$ROOT" python3 - "$TMP" <<'PY'
import os, struct, sys
import numpy as np
sys.path.insert(0, os.environ["DMND_ROOT"])
from diamond_amd import synth
raw = open(os.path.join(os.environ["DMND_ROOT"], "diamond_amd", "motifs.bin"), "rb").read()
codes = struct.unpack("<%dQ" % (len(raw) // 8), raw)
rng = np.random.default_rng(12)
def letters(code):
    out = []
    for _ in range(8):
        out.append(code % 20); code //= 20
    return np.array(out[::-1], np.int8)
db, doff, q, qoff = synth.generate(120, members=6, queries=160, seed=21)
def plant(data, off):
    data = data.copy()
    for i in range(len(off) - 1):
        b, e = int(off[i]), int(off[i + 1])
        r = rng.random()
        if r < 0.5:
            for _ in range(int(rng.integers(1, 4))):
                p = int(rng.integers(b, max(b + 1, e - 8)))
                data[p:p + 8] = letters(codes[int(rng.integers(0, len(codes)))])[:max(0, min(8, e - p))]
        elif r < 0.6 and e - b > 60:                       # five motifs back to back: a 40-letter range, too long to mask
            p = int(rng.integers(b, e - 41))
            for k in range(5):
                data[p + 8 * k:p + 8 * k + 8] = letters(codes[int(rng.integers(0, len(codes)))])
        elif r < 0.65:                                     # a short sequence that is mostly motifs: not masked at all
            n = min(e - b, 24)
            for k in range(n // 8):
                data[b + 8 * k:b + 8 * k + 8] = letters(codes[int(rng.integers(0, len(codes)))])
    return data
synth.write_fasta(sys.argv[1] + "/db.faa", "t", plant(db, doff), doff)
synth.write_fasta(sys.argv[1] + "/q.faa", "q", plant(q, qoff), qoff)
PY
cp "$TMP/db.faa" "$HERE/motif_db.faa"; cp "$TMP/q.faa" "$HERE/motif_q.faa"
DIAMOND_TAP_EXT="$HERE/ext_motif.tap" "$TAP" blastp --algo 0 -q "$TMP/q.faa" -d "$TMP/db.faa" -o "$HERE/motif.tsv" -p2 2>/dev/null
DIAMOND_TAP_EXT="$HERE/ext_motif_a1.tap" "$TAP" blastp --fast --algo 1 -q "$TMP/q.faa" -d "$TMP/db.faa" -o "$HERE/motif_a1.tsv" -p2 2>/dev/null
"$TAP" blastp --algo 0 --motif-masking 0 -q "$TMP/q.faa" -d "$TMP/db.faa" -o "$TMP/off.tsv" -p2 2>/dev/null
echo "lines with motif masking: $(wc -l < "$HERE/motif.tsv"), without: $(wc -l < "$TMP/off.tsv"), differing: $(diff <(sort "$HERE/motif.tsv") <(sort "$TMP/off.tsv") | grep -c '^[<>]' || true)"
ls -la "$HERE"/ext_motif*.tap
rm -rf "$TMP"
