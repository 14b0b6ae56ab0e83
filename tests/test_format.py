"""Output formats on the CPU (host-only part of the C ABI): every line of the reference's `-f 6 FIELD...` output (37 fields,
tests/golden/fields_k4.tsv.gz: reference run on its own test fixture with -k 4) is rebuilt from the record + packed transcript
that the line itself describes and must come out byte-identical; the same records give the reference's `-f 0` and `-f paf`
files (tests/golden/pairwise_k4.out.gz, paf_k4.out.gz). Mirrors src/output/blast_tab_format.cpp, blast_pairwise_format.cpp,
paf_format.cpp."""
import ctypes
import gzip
import os

import numpy as np
import pytest

from diamond_amd import hip

HERE = os.path.dirname(os.path.abspath(__file__))
FIELDS = ["qseqid", "qlen", "sseqid", "sallseqid", "slen", "qstart", "qend", "sstart", "send", "qseq", "sseq", "evalue", "bitscore", "score",
          "length", "pident", "nident", "mismatch", "positive", "gapopen", "gaps", "ppos", "qframe", "btop", "stitle", "salltitles", "qcovhsp",
          "qtitle", "full_sseq", "qnum", "snum", "scovhsp", "full_qseq", "qseq_gapped", "sseq_gapped", "qstrand", "cigar"]
AA = "ARNDCQEGHILKMFPSTWYVBJZX*_"
CODE = {c: i for i, c in enumerate(AA)}


def letters(s):
    return np.array([CODE[c] for c in s], dtype=np.int8)


def transcript_of(qg, sg):
    """PackedOperation bytes from the two gapped strings (basic/packed_transcript.h)."""
    out = []
    for a, b in zip(qg, sg):
        if a == "-":
            out.append((2 << 6) | CODE[b])
        elif b == "-":
            if out and out[-1] >> 6 == 1 and (out[-1] & 63) < 63:
                out[-1] += 1
            else:
                out.append((1 << 6) | 1)
        elif a == b:
            if out and out[-1] >> 6 == 0 and (out[-1] & 63) < 63:
                out[-1] += 1
            else:
                out.append(1)
        else:
            out.append((3 << 6) | CODE[b])
    return np.array(out, dtype=np.uint8)


def records():
    lib = hip.load()
    lib.dmnd_bitscore_p.restype = ctypes.c_double
    lib.dmnd_bitscore_p.argtypes = [ctypes.c_void_p, ctypes.c_double]
    params = hip.default_params()
    for line in gzip.open(os.path.join(HERE, "golden", "fields_k4.tsv.gz"), "rt"):
        f = dict(zip(FIELDS, line.rstrip("\n").split("\t")))
        m = np.zeros(1, dtype=hip.MATCH_DTYPE)[0]
        m["query"], m["target"] = int(f["qnum"]), int(f["snum"])
        # the record carries the exact bit score (PAF prints it truncated, the tabular formats rounded)
        m["evalue"], m["bit_score"] = float(f["evalue"]), lib.dmnd_bitscore_p(ctypes.byref(params), float(f["score"]))
        h = m["hsp"]
        h["score"], h["q_begin"], h["q_end"], h["s_begin"], h["s_end"] = int(f["score"]), int(f["qstart"]) - 1, int(f["qend"]), int(f["sstart"]) - 1, int(f["send"])
        h["length"], h["identities"], h["mismatches"], h["positives"] = int(f["length"]), int(f["nident"]), int(f["mismatch"]), int(f["positive"])
        h["gap_openings"], h["gaps"] = int(f["gapopen"]), int(f["gaps"])
        tr = transcript_of(f["qseq_gapped"], f["sseq_gapped"])
        h["transcript_len"] = len(tr)
        yield line, f, m, tr


def test_every_field_of_the_reference_output_is_reproduced():
    ids, need = hip.output_fields(FIELDS)
    assert need and len(ids) == len(FIELDS)
    n = 0
    for line, f, m, tr in records():
        got = hip.format_fields(ids, m, tr, f["qtitle"], f["stitle"], letters(f["full_qseq"]), int(f["slen"]), full_sseq=letters(f["full_sseq"]),
                                qnum=int(f["qnum"]), snum=int(f["snum"]))
        assert got == line, (n, [(k, a, b) for k, a, b in zip(FIELDS, line.split("\t"), got.split("\t")) if a != b][:3])
        n += 1
    assert n > 600


def test_pairwise_paf_and_sam_files_are_reproduced():
    p = hip.default_params()
    M = np.array(p.matrix8, dtype=np.int8)
    pw, paf, sam, last = ["BLASTP 2.3.0+\n\n\n"], [], [], None
    for line, f, m, tr in records():
        if f["qtitle"] != last:
            pw.append(hip.format_pairwise_intro(f["qtitle"], int(f["qlen"])))
            last = f["qtitle"]
        q = letters(f["full_qseq"])
        pw.append(hip.format_pairwise(m, tr, f["qtitle"], f["stitle"], q, int(f["slen"]), M))
        paf.append(hip.format_paf(m, f["qtitle"], f["stitle"], q, int(f["slen"])))
        sam.append(hip.format_sam(m, tr, f["qtitle"], f["stitle"], q, int(f["slen"])))
    assert "".join(pw) == gzip.open(os.path.join(HERE, "golden", "pairwise_k4.out.gz"), "rt").read()
    assert "".join(paf) == gzip.open(os.path.join(HERE, "golden", "paf_k4.out.gz"), "rt").read()
    assert "".join(sam) == gzip.open(os.path.join(HERE, "golden", "sam_k4.body.gz"), "rt").read()


def test_field_names_are_checked_like_the_reference():
    with pytest.raises(hip.DiamondHipError, match="Invalid output field: nosuchfield"):
        hip.output_fields(["qseqid", "nosuchfield"])
    with pytest.raises(hip.DiamondHipError, match="not available in this build"):
        hip.output_fields(["staxids"])
    ids, need = hip.output_fields(["qseqid", "sseqid", "pident"])
    assert not need
    # a field that reads the transcript fails loudly when the extension ran without an arena
    m = next(records())
    ids, _ = hip.output_fields(["cigar"])
    with pytest.raises(hip.DiamondHipError, match="needs the transcript"):
        hip.format_fields(ids, m[2], None, "q", "s", letters(m[1]["full_qseq"]), int(m[1]["slen"]))
    assert hip.format_pairwise_intro("q7 some title", 120, unaligned=True) == "Query= q7 some title\n\nLength=120\n\n\n***** No hits found *****\n\n\n"
