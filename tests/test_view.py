"""`diamond-hip view` (host only): a DAA archive printed in the other formats must read as the reference's own `view` prints it
(tests/golden/view_golden.txt.gz, minted by tests/golden/make_view_golden.sh) -- a blastx archive written by the reference (DNA
coordinates, frames, the e-value rule of view) and the blastp archive of the ctest fixture; plus the legacy `-a` option and damaged
files."""
import gzip
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "diamond_amd", "diamond-hip")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _sections():
    out, name = {}, None
    for line in gzip.open(os.path.join(GOLDEN, "view_golden.txt.gz"), "rt"):
        if line.startswith("#### "):
            name = line[5:].rstrip("\n")
            out[name] = []
        else:
            out[name].append(line)
    return {k: "".join(v) for k, v in out.items()}


@pytest.fixture(scope="module")
def archives(tmp_path_factory):
    d = tmp_path_factory.mktemp("daa")
    for a, src in (("bx", "daa_blastx.daa.gz"), ("k4", "daa_k4.daa.gz")):
        open(d / (a + ".daa"), "wb").write(gzip.open(os.path.join(GOLDEN, src), "rb").read())
    return d


def test_view_prints_what_the_reference_view_prints(archives):
    assert os.path.exists(CLI), "diamond-hip not built (make product)"
    sections = _sections()
    assert len(sections) == 12
    for name, want in sections.items():
        a, *args = name.split()
        r = subprocess.run([CLI, "view", "-a", str(archives / (a + ".daa")), "-o", str(archives / "out")] + args, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (name, r.stderr)
        got = open(archives / "out").read()
        if args[:2] == ["-f", "5"]:
            got, want = "".join(l for l in got.splitlines(True) if "<BlastOutput_version>" not in l).rstrip("\n"), want.rstrip("\n")
        if args[:2] == ["-f", "101"]:
            got = "".join(l for l in got.splitlines(True) if not l.startswith("@PG"))
        assert got == want, name
        assert len(want) > 1000, name


def test_view_options_and_damaged_files(archives):
    r = subprocess.run([CLI, "view", "-a", str(archives / "bx"), "-f", "6", "qseqid", "full_sseq"], capture_output=True, text=True)      # ".daa" is appended
    assert r.returncode != 0 and "not stored in a DAA file" in r.stderr
    r = subprocess.run([CLI, "view", "-o", "x"], capture_output=True, text=True)
    assert r.returncode != 0 and "requires a DAA" in r.stderr
    raw = open(archives / "k4.daa", "rb").read()
    open(archives / "bad1.daa", "wb").write(b"\0" * 8 + raw[8:])
    open(archives / "bad2.daa", "wb").write(raw[: len(raw) // 2])
    open(archives / "bad3.daa", "wb").write(raw[:16 + 128] + b"\0" * 8 + raw[16 + 136:])          # block size 0: run did not finish
    for name, msg in (("bad1", "not a DAA file"), ("bad2", "Truncated DAA file"), ("bad3", "has probably not completed")):
        r = subprocess.run([CLI, "view", "-a", str(archives / (name + ".daa")), "-o", str(archives / "x")], capture_output=True, text=True)
        assert r.returncode != 0 and msg in r.stderr, (name, r.stderr)
    # gzip output of view
    r = subprocess.run([CLI, "view", "-a", str(archives / "bx.daa"), "--compress", "1", "-o", str(archives / "z.tsv")], capture_output=True, text=True)
    assert r.returncode == 0 and gzip.open(str(archives / "z.tsv.gz"), "rt").read() == _sections()["bx -f 6"]


def test_view_of_frameshift_alignments(tmp_path):
    """A DAA archive the reference wrote with `blastx -F 15` (154 of its 157 alignments change frame): every field that walks the
    alignment -- btop, cigar, qseq_gapped, sseq_gapped, sseq -- and the statistics and read coordinates recomputed from the
    transcripts must read as the reference's own `view` prints them (round 4: the formatter's cursor follows the three frames).
    qseq_translated is left out: with a frame change the reference's view reads past the end of the first frame."""
    fields = "qseqid sseqid pident length mismatch gapopen qstart qend sstart send evalue bitscore qframe nident positive gaps btop cigar qseq_gapped sseq_gapped sseq qcovhsp".split()
    open(tmp_path / "fs.daa", "wb").write(gzip.open(os.path.join(GOLDEN, "fs_f15.daa.gz"), "rb").read())
    r = subprocess.run([CLI, "view", "-a", str(tmp_path / "fs.daa"), "-o", str(tmp_path / "out"), "-f", "6"] + fields, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    want = gzip.open(os.path.join(GOLDEN, "fs_f15_view_fields.tsv.gz"), "rt").read()
    got = open(tmp_path / "out").read()
    assert want.count("\\") + want.count("/") > 400            # the frame shifts are there
    if got != want:
        for i, (a, b) in enumerate(zip(want.splitlines(), got.splitlines())):
            assert a == b, (i, [(x, y) for x, y in zip(a.split("\t"), b.split("\t")) if x != y][:2])
    assert got == want


def test_view_of_frameshift_alignments_in_the_other_formats(tmp_path):
    """The same archive in the pairwise, XML, PAF and SAM formats: the coordinates of the pairwise lines move by single bases where
    the alignment changes frame, the midline and MD:Z skip the shift columns. One SAM line is left out: for r75 / t326 the reference
    prints the letter one past the end of the first frame (its query-sequence column reads the range of the walk from ONE frame)."""
    open(tmp_path / "fs.daa", "wb").write(gzip.open(os.path.join(GOLDEN, "fs_f15.daa.gz"), "rb").read())
    sections, name = {}, None
    for line in gzip.open(os.path.join(GOLDEN, "fs_f15_view_formats.txt.gz"), "rt"):
        if line.startswith("#### "):
            name = line[5:].rstrip("\n")
            sections[name] = []
        else:
            sections[name].append(line)
    assert sorted(sections) == ["-f 0", "-f 101", "-f 5", "-f paf"]
    drop = lambda t: "".join(l for l in t.splitlines(True) if "<BlastOutput_version>" not in l and not l.startswith("@PG") and not l.startswith("r75\t0\tt326\t"))
    for name, lines in sections.items():
        r = subprocess.run([CLI, "view", "-a", str(tmp_path / "fs.daa"), "-o", str(tmp_path / "out")] + name.split(), capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (name, r.stderr)
        want, got = drop("".join(lines)).rstrip("\n"), drop(open(tmp_path / "out", errors="replace").read()).rstrip("\n")
        assert len(want) > 10000, name
        if got != want:
            for i, (a, b) in enumerate(zip(want.splitlines(), got.splitlines())):
                assert a == b, (name, i, a[:200], b[:200])
        assert got == want, name


def test_view_writes_every_daa_record_back(tmp_path, archives):
    """DMND_VIEW_CHECK_DAA=1: every match record that `view` reads is written again from the rebuilt record (dmnd_format_daa_match) and
    must be the bytes that were read -- for the blastp and blastx archives and for the archive of a frameshift run (the record of a
    frameshift alignment is written from its first column's frame and position)."""
    open(tmp_path / "fs.daa", "wb").write(gzip.open(os.path.join(GOLDEN, "fs_f15.daa.gz"), "rb").read())
    for path in (tmp_path / "fs.daa", archives / "bx.daa", archives / "k4.daa"):
        r = subprocess.run([CLI, "view", "-a", str(path), "-o", str(tmp_path / "out")], capture_output=True, text=True, timeout=120, env=dict(os.environ, DMND_VIEW_CHECK_DAA="1"))
        assert r.returncode == 0 and "written back: 0 differ" in r.stderr, (path, r.stderr[-300:])
