"""Input side of the command line that needs no device: `diamond-hip makedb` from FASTA, gzip-compressed FASTA, FASTQ and
gzip-compressed FASTQ writes the same .dmnd; malformed inputs fail with the reference's messages (data/fasta/fasta_file.cpp:43-52,
data/fasta/parser.h:238-270)."""
import gzip
import os
import subprocess

import pytest

from diamond_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "diamond_amd", "diamond-hip")


def _makedb(src, dst):
    return subprocess.run([CLI, "makedb", "--in", str(src), "-d", str(dst)], capture_output=True, text=True, timeout=120)


def test_makedb_reads_gzip_and_fastq(tmp_path):
    assert os.path.exists(CLI), "diamond-hip not built (make product)"
    db, doff, _, _ = synth.generate(20, members=5, queries=1, seed=3)
    synth.write_fasta(str(tmp_path / "a.faa"), "t", db, doff)
    recs = open(tmp_path / "a.faa").read().split(">")[1:]
    with open(tmp_path / "a.fastq", "w") as f:
        for i, r in enumerate(recs):
            h, *s = r.strip().split("\n")
            s = "".join(s)
            half = len(s) // 2
            if i % 3 == 0:            # sequence and quality wrapped over two lines, '+' line repeating the title, CRLF line ends
                f.write("@%s\r\n%s\r\n%s\r\n+%s\r\n%s\r\n%s\r\n" % (h, s[:half], s[half:], h, "@" * half, "@" * (len(s) - half)))
            else:
                f.write("@%s\n%s\n+\n%s\n" % (h, s, "@" * len(s)))      # quality lines that start with '@'
    for name in ("a.faa", "a.fastq"):
        with open(tmp_path / name, "rb") as f, gzip.open(str(tmp_path / name) + ".gz", "wb") as g:
            g.write(f.read())
    out = []
    for name in ("a.faa", "a.faa.gz", "a.fastq", "a.fastq.gz"):
        r = _makedb(tmp_path / name, tmp_path / ("db_" + name))
        assert r.returncode == 0, r.stderr
        out.append(open(str(tmp_path / ("db_" + name)) + ".dmnd", "rb").read())
    assert len(out[0]) > 1000 and all(o == out[0] for o in out)


@pytest.mark.parametrize("content, message", [
    (b"", "Input file seems to be empty"),
    (b"ACDEFG\n", "First line must begin with '>' (FASTA) or '@' (FASTQ)"),
    (b"@r1\nACDE\n", "Malformed FASTQ record at line 1"),
    (b"@r1\nACDE\n+\nIIII\nr2\nACDE\n+\nIIII\n", "Malformed FASTQ record at line 5"),
    (b">s1\nAC1DE\n", "Invalid character (1) in sequence s1"),
])
def test_makedb_rejects_malformed_input(tmp_path, content, message):
    open(tmp_path / "bad", "wb").write(content)
    r = _makedb(tmp_path / "bad", tmp_path / "bad_db")
    assert r.returncode != 0 and message in r.stderr, r.stderr


@pytest.mark.parametrize("args, message", [
    (["--max-hsps", "-2"], "Invalid value for --max-hsps"),
    (["--top", "10", "-k", "5"], "--top and -k/--max-target-seqs are mutually exclusive"),
    (["-F", "15"], "Frameshift alignments are only supported for translated searches."),
    (["--comp-based-stats", "7"], "Invalid value for --comp-based-stats. Permitted values: 0, 1, 2, 3, 4, 5."),
    (["--custom-matrix", "m.txt"], "--custom-matrix is not part of this build"),
    (["--iterate"], "--iterate is not part of this build"),
    (["--bogus-option"], "Invalid option: --bogus-option"),
    (["--strand", "sideways"], "Invalid value for parameter --strand"),
    (["--compress", "zstd"], "not compiled with ZStd"),
    (["--ext", "global"], "is not part of this build"),
    (["--unfmt", "fastq"], "Only the fasta format"),
])
def test_unsupported_options_fail_with_a_message(args, message):
    r = subprocess.run([CLI, "blastp", "-q", "x", "-d", "y", "--tmpdir", "/tmp", "--max-hsps", "1", "--ignore-warnings"] + args, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and message in r.stderr, r.stderr


def test_makedb_reads_standard_input(tmp_path):
    """No --in (or "-"): the sequences come from standard input, plain or gzip-compressed (File::Flags::TREAT_BLANK_AS_STDIN)."""
    db, doff, _, _ = synth.generate(10, members=5, queries=1, seed=4)
    synth.write_fasta(str(tmp_path / "a.faa"), "t", db, doff)
    raw = open(tmp_path / "a.faa", "rb").read()
    assert _makedb(tmp_path / "a.faa", tmp_path / "file").returncode == 0
    want = open(tmp_path / "file.dmnd", "rb").read()
    for name, data, args in (("plain", raw, []), ("gz", gzip.compress(raw), []), ("dash", raw, ["--in", "-"])):
        r = subprocess.run([CLI, "makedb", "-d", str(tmp_path / name)] + args, input=data, capture_output=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert open(str(tmp_path / name) + ".dmnd", "rb").read() == want, name
    r = subprocess.run([CLI, "makedb", "-d", str(tmp_path / "none")], input=b"", capture_output=True, timeout=120)
    assert r.returncode != 0 and b"seems to be empty" in r.stderr


def test_dbinfo_prints_the_header_like_the_reference(tmp_path):
    db, doff, _, _ = synth.generate(12, members=5, queries=1, seed=6)
    synth.write_fasta(str(tmp_path / "a.faa"), "t", db, doff)
    assert _makedb(tmp_path / "a.faa", tmp_path / "a").returncode == 0
    r = subprocess.run([CLI, "dbinfo", "-d", str(tmp_path / "a")], capture_output=True, text=True, timeout=60)
    want = ("          Database type  Diamond database\nDatabase format version  3\n          Diamond build  182\n"
            "              Sequences  %d\n                Letters  %d\n" % (len(doff) - 1, int(doff[-1])))
    assert r.returncode == 0 and r.stdout == want
    ref = os.path.join(ROOT, "oracle", "_ref", "diamond")
    if os.path.exists(ref):
        assert subprocess.run([ref, "dbinfo", "-d", str(tmp_path / "a.dmnd"), "--quiet"], capture_output=True, text=True, timeout=60).stdout == want
    r = subprocess.run([CLI, "dbinfo", "-d", str(tmp_path / "a.faa")], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "not a DIAMOND database" in r.stderr
