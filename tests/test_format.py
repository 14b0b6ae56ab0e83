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


def transcript_as_traced(qg, sg):
    """The transcript bytes as the traceback leaves them (Hsp::push_match / push_gap, basic/hssp.cpp:260-290): one byte per match
    column, an insertion as runs of at most 63 pushed from the alignment's end (so the remainder comes first after the reversal)."""
    out, i = [], 0
    while i < len(qg):
        a, b = qg[i], sg[i]
        if b == "-":
            j = i
            while j < len(qg) and sg[j] == "-":
                j += 1
            n, runs = j - i, []
            while n > 0:
                runs.append(min(n, 63))
                n -= runs[-1]
            out.extend((1 << 6) | r for r in reversed(runs))
            i = j
            continue
        out.append((2 << 6) | CODE[b] if a == "-" else 1 if a == b else (3 << 6) | CODE[b])
        i += 1
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


def test_xml_file_is_reproduced():
    """`-f 5` (tests/golden/xml_k4.out.gz): header, one <Iteration> per query with its <Hit> / <Hsp> elements, statistics block, footer."""
    p = hip.default_params()
    M = np.array(p.matrix8, dtype=np.int8)
    recs = list(records())
    n_db = len({f["stitle"] for _, f, _, _ in recs} | {f["qtitle"] for _, f, _, _ in recs})
    fasta = open(os.path.join(HERE, "golden", "ref_ctest", "data.faa")).read().split(">")[1:]
    db_letters = sum(len("".join(r.split("\n")[1:])) for r in fasta)
    assert n_db <= len(fasta)
    first = fasta[0].split("\n")
    out = [hip.format_xml_header("blastp", "diamond 2.2.2", "data.faa", first[0], len("".join(first[1:])))]
    last, hit = None, 0
    for line, f, m, tr in recs:
        if f["qtitle"] != last:
            if last is not None:
                out.append(hip.format_xml_query_epilog(False, len(fasta), db_letters, p.K, p.lambda_))
            out.append(hip.format_xml_query_intro(f["qtitle"], int(f["qnum"]), int(f["qlen"])))
            last, hit = f["qtitle"], 0
        out.append(hip.format_xml(m, tr, f["qtitle"], f["stitle"], letters(f["full_qseq"]), int(f["slen"]), hit, 0, M))
        hit += 1
    out.append(hip.format_xml_query_epilog(False, len(fasta), db_letters, p.K, p.lambda_))
    out.append(hip.XML_FOOTER)
    want = gzip.open(os.path.join(HERE, "golden", "xml_k4.out.gz"), "rt").read()
    got = "".join(out)
    if got != want:
        g, w = got.splitlines(), want.splitlines()
        bad = next(i for i in range(min(len(g), len(w))) if g[i] != w[i])
        raise AssertionError("line %d:\n  got  %s\n  want %s" % (bad + 1, g[bad][:200], w[bad][:200]))
    # titles: escapes, several titles of one record, accession parsing (Util::Seq::get_accession)
    line, f, m, tr = recs[0]
    x = hip.format_xml(m, tr, "q", 'gi|123|ref|NP_000001.2| first <protein> & "more"\x01sp|P12345|NAME_HUMAN second', letters(f["full_qseq"]), int(f["slen"]), 2, 0, M)
    assert x.startswith("  </Hit_hsps>\n</Hit>\n<Hit>\n  <Hit_num>3</Hit_num>\n  <Hit_id>gi|123|ref|NP_000001.2|</Hit_id>\n"
                        "  <Hit_def>first &lt;protein&gt; &amp; &quot;more&quot; &gt;sp|P12345|NAME_HUMAN second</Hit_def>\n  <Hit_accession>NP_000001</Hit_accession>\n")
    assert "<Hit_accession>Q6GZX4</Hit_accession>" in hip.format_xml(m, tr, "q", "UniRef90_Q6GZX4 x", letters(f["full_qseq"]), int(f["slen"]), 0, 0, M)
    assert hip.format_xml(m, tr, "q", "t", letters(f["full_qseq"]), int(f["slen"]), 0, 1, M).startswith("    <Hsp>\n      <Hsp_num>2</Hsp_num>")
    assert "<Iteration_query-def>a &apos;b&apos;</Iteration_query-def>" in hip.format_xml_query_intro("a 'b'\x01second title", 4, 10)
    assert hip.format_xml_query_epilog(True, -1, -1, 0.041, 0.267).startswith("</Iteration_hits>\n  <Iteration_stat>\n    <Statistics>\n      <Statistics_hsp-len>")


def test_daa_file_is_reproduced():
    """`-f 100` (tests/golden/daa_k4.daa.gz, reference run with -p 1): headers, query records with packed sequences, match records with
    their packed coordinates and transcripts, dictionary of the targets in order of first appearance, lengths -- byte for byte."""
    import struct
    p = hip.default_params()
    recs = list(records())
    fasta = open(os.path.join(HERE, "golden", "ref_ctest", "data.faa")).read().split(">")[1:]
    db_letters = sum(len("".join(r.split("\n")[1:])) for r in fasta)
    head = dict(build=182, db_seqs=len(fasta), db_letters=db_letters, mode=2, gap_open=11, gap_extend=1, K=p.K, lambda_=p.lambda_, max_evalue=0.001, matrix="BLOSUM62")
    body, dict_ids, names, lens = [], {}, [], []
    last, rec, n_queries = None, None, 0

    def close(r):
        if r is not None:
            body.append(struct.pack("<I", len(r) - 4) + r[4:])

    for line, f, m, tr in recs:
        if f["qtitle"] != last:
            close(rec)
            rec = hip.format_daa_query(f["qtitle"], letters(f["full_qseq"]))
            last = f["qtitle"]
            n_queries += 1
        key = f["stitle"]
        if key not in dict_ids:
            dict_ids[key] = len(names)
            names.append(f["sseqid"])
            lens.append(int(f["slen"]))
        raw = transcript_as_traced(f["qseq_gapped"], f["sseq_gapped"])
        m = m.copy()
        m["hsp"]["transcript_len"] = len(raw)
        rec += hip.format_daa_match(m, raw, f["qtitle"], f["stitle"], letters(f["full_qseq"]), int(f["slen"]), dict_ids[key])
    close(rec)
    body.append(struct.pack("<I", 0))
    aln = b"".join(body)
    ref_names = b"".join(n.encode() + b"\0" for n in names)
    header = hip.format_daa_header(db_seqs_used=len(names), query_records=n_queries, finished=1, alignment_bytes=len(aln), ref_name_bytes=len(ref_names), **head)
    assert len(header) == 2448 and len(hip.format_daa_header(**head)) == 2448
    got = header + aln + ref_names + b"".join(struct.pack("<I", x) for x in lens)
    want = gzip.open(os.path.join(HERE, "golden", "daa_k4.daa.gz"), "rb").read()
    if got != want:
        bad = next(i for i in range(min(len(got), len(want))) if got[i] != want[i])
        raise AssertionError("first difference at byte %d of %d / %d: got %r want %r" % (bad, len(got), len(want), got[bad:bad + 24], want[bad:bad + 24]))
    # a translated query: DNA packed with 2 bits per base, 3 when it holds an N; reverse frames set bit 6 of the match flag
    assert hip.format_daa_query("r1 x", np.array([0, 1, 2, 3, 3, 2, 1, 0], np.int8), dna=True) == struct.pack("<II", 0, 8) + b"r1\0" + bytes([0]) + bytes([0xE4, 0x1B])
    assert hip.format_daa_query("r1", np.array([0, 4, 2], np.int8), dna=True)[11] == 1


def test_format_switches_of_the_xml_and_sam_writers():
    """--xml-blord-format, --no-parse-seqids, --sam-query-len (dmnd_set_format_flags; checked against the reference binary's output for
    the same options while they were written: tests/golden/make_format_golden.sh documents the commands)."""
    p = hip.default_params()
    M = np.array(p.matrix8, dtype=np.int8)
    line, f, m, tr = next(records())
    q = letters(f["full_qseq"])
    title = "gi|123|ref|NP_000001.2| first protein\x01sp|P12345|NAME_HUMAN second"
    try:
        plain = hip.format_xml(m, tr, "q", title, q, int(f["slen"]), 0, 0, M, snum=7)
        assert "<Hit_id>gi|123|ref|NP_000001.2|</Hit_id>" in plain and "<Hit_accession>NP_000001</Hit_accession>" in plain
        hip.set_format_flags(hip.FMT_XML_BLORD)
        x = hip.format_xml(m, tr, "q", title, q, int(f["slen"]), 0, 0, M, snum=7)
        assert "  <Hit_id>gnl|BL_ORD_ID|7</Hit_id>\n  <Hit_def>gi|123|ref|NP_000001.2| first protein &gt;sp|P12345|NAME_HUMAN second</Hit_def>\n  <Hit_accession>NP_000001</Hit_accession>" in x
        hip.set_format_flags(hip.FMT_NO_PARSE_SEQIDS)
        x = hip.format_xml(m, tr, "q", title, q, int(f["slen"]), 0, 0, M, snum=7)
        assert "<Hit_id>gi|123|ref|NP_000001.2|</Hit_id>" in x and "<Hit_accession>gi|123|ref|NP_000001.2|</Hit_accession>" in x
        sam = hip.format_sam(m, tr, f["qtitle"], f["stitle"], q, int(f["slen"]))
        assert "ZQ:i:" not in sam
        hip.set_format_flags(hip.FMT_SAM_QUERY_LEN)
        assert hip.format_sam(m, tr, f["qtitle"], f["stitle"], q, int(f["slen"])) == sam[:-1] + "\tZQ:i:%d\n" % len(q)
        with pytest.raises(hip.DiamondHipError, match="unknown flag"):
            hip.set_format_flags(64)
    finally:
        hip.set_format_flags(0)


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
