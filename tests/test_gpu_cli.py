"""-m gpu test of the command line surface: `diamond-hip makedb` + `diamond-hip blastp --fast` (C++ host over the C ABI,
all alignment work on the MI355X) against the unmodified reference binary on the same files: the .dmnd written by our
makedb must be readable by the reference, and the tabular outputs must be byte-identical."""
import os
import subprocess
import numpy as np
import pytest

from diamond_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "diamond")
CLI = os.path.join(ROOT, "diamond_amd", "diamond-hip")


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r


def test_cli_makedb_blastp_matches_reference(tmp_path):
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    assert os.path.exists(CLI), "diamond-hip not built (make product)"
    db, doff, q, qoff = synth.generate(400, members=10, queries=500, seed=5)
    synth.write_fasta(str(tmp_path / "db.faa"), "t", db, doff)
    synth.write_fasta(str(tmp_path / "q.faa"), "q", q, qoff)
    _run([CLI, "makedb", "--in", str(tmp_path / "db.faa"), "-d", str(tmp_path / "db")])
    ref_args = ["blastp", "--fast", "--algo", "0", "--masking", "0", "--motif-masking", "0", "-q", str(tmp_path / "q.faa"), "-p", "4"]
    _run([REF] + ref_args + ["-d", str(tmp_path / "db.dmnd"), "-o", str(tmp_path / "ref.tsv")])          # our .dmnd, reference reader
    _run([CLI, "blastp", "--fast", "--masking", "0", "-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.dmnd"), "-o", str(tmp_path / "hip.tsv"), "-p", "4"])
    _run([CLI, "blastp", "--fast", "--masking", "0", "-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.faa"), "-o", str(tmp_path / "hip2.tsv"), "-p", "4", "-k", "3", "-e", "1e-10"])
    _run([REF] + ref_args + ["-d", str(tmp_path / "db.faa"), "-o", str(tmp_path / "ref2.tsv"), "-k", "3", "-e", "1e-10"])
    ref = open(tmp_path / "ref.tsv").read()
    assert len(ref.splitlines()) > 300
    assert open(tmp_path / "hip.tsv").read() == ref
    assert open(tmp_path / "hip2.tsv").read() == open(tmp_path / "ref2.tsv").read()
    # no composition based statistics (reference ctest diamond-test-blastp-comp-based-stats-0)
    _run([REF] + [a for a in ref_args if a != "--fast"] + ["--comp-based-stats", "0", "-d", str(tmp_path / "db.dmnd"), "-o", str(tmp_path / "ref5.tsv")])
    _run([CLI, "blastp", "--masking", "0", "--comp-based-stats", "0", "-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.dmnd"), "-o", str(tmp_path / "hip5.tsv"), "-p", "4"])
    assert open(tmp_path / "hip5.tsv").read() == open(tmp_path / "ref5.tsv").read()
    # default sensitivity (two shapes + stage-2 ungapped e-value filter)
    _run([REF] + [a for a in ref_args if a != "--fast"] + ["-d", str(tmp_path / "db.dmnd"), "-o", str(tmp_path / "ref3.tsv")])
    _run([CLI, "blastp", "--masking", "0", "-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.dmnd"), "-o", str(tmp_path / "hip3.tsv"), "-p", "4"])
    ref3 = open(tmp_path / "ref3.tsv").read()
    assert len(ref3.splitlines()) >= len(ref.splitlines())
    assert open(tmp_path / "hip3.tsv").read() == ref3
    # --sensitive (16 shapes, ungapped + gapped filters)
    _run([REF] + [a if a != "--fast" else "--sensitive" for a in ref_args] + ["-d", str(tmp_path / "db.dmnd"), "-o", str(tmp_path / "ref4.tsv")])
    _run([CLI, "blastp", "--sensitive", "--masking", "0", "-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.dmnd"), "-o", str(tmp_path / "hip4.tsv"), "-p", "4"])
    ref4 = open(tmp_path / "ref4.tsv").read()
    assert len(ref4.splitlines()) >= len(ref3.splitlines())
    assert open(tmp_path / "hip4.tsv").read() == ref4


def test_cli_blastx_matches_reference(tmp_path):
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    db, doff, q, qoff = synth.generate(300, members=10, queries=250, seed=11)
    dna, off = synth.back_translate(q, qoff, seed=12)
    synth.write_fasta(str(tmp_path / "db.faa"), "t", db, doff)
    synth.write_dna_fasta(str(tmp_path / "reads.fna"), "r", dna, off)
    for mode in ([], ["--fast"], ["--sensitive"]):
        tag = (mode or ["default"])[0].strip("-")
        _run([REF, "blastx"] + mode + ["--algo", "0", "--masking", "0", "--motif-masking", "0", "-q", str(tmp_path / "reads.fna"),
                                        "-d", str(tmp_path / "db.faa"), "-o", str(tmp_path / ("ref_%s.tsv" % tag)), "-p", "4"])
        _run([CLI, "blastx"] + mode + ["--masking", "0", "-q", str(tmp_path / "reads.fna"), "-d", str(tmp_path / "db.faa"),
                                        "-o", str(tmp_path / ("hip_%s.tsv" % tag)), "-p", "4"])
        ref = open(tmp_path / ("ref_%s.tsv" % tag)).read()
        assert len(ref.splitlines()) > 150
        assert open(tmp_path / ("hip_%s.tsv" % tag)).read() == ref, tag


def _plant_repeats(data, off, rng, frac=0.3):
    """Low-complexity stretches (tandem repeats with a few substitutions) inside a fraction of the sequences."""
    data = data.copy()
    for i in range(len(off) - 1):
        n = int(off[i + 1] - off[i])
        if n < 80 or rng.random() > frac:
            continue
        unit = rng.integers(0, 20, int(rng.integers(1, 7))).astype(data.dtype)
        a = int(rng.integers(0, n - 60))
        L = int(rng.integers(25, 60))
        seg = np.resize(unit, L)
        flip = rng.random(L) < 0.05
        seg[flip] = rng.integers(0, 20, int(flip.sum()))
        data[off[i] + a: off[i] + a + L] = seg
    return data


def test_cli_default_masking_matches_reference(tmp_path):
    """Default --masking (tantan on both blocks, on the GPU) against the reference's default masking; motif masking off."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    rng = np.random.default_rng(21)
    db, doff, q, qoff = synth.generate(300, members=10, queries=400, seed=21)
    db, q = _plant_repeats(db, doff, rng), _plant_repeats(q, qoff, rng)
    synth.write_fasta(str(tmp_path / "db.faa"), "t", db, doff)
    synth.write_fasta(str(tmp_path / "q.faa"), "q", q, qoff)
    for mode in ([], ["--fast"], ["--mid-sensitive"], ["--sensitive"], ["--more-sensitive"], ["--very-sensitive"]):
        tag = (mode or ["default"])[0].strip("-")
        _run([REF, "blastp"] + mode + ["--algo", "0", "--motif-masking", "0", "-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.faa"),
                                        "-o", str(tmp_path / ("ref_%s.tsv" % tag)), "-p", "4"])
        r = _run([CLI, "blastp"] + mode + ["--motif-masking", "0", "-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.faa"),
                                            "-o", str(tmp_path / ("hip_%s.tsv" % tag)), "-p", "4"])
        assert "masked letters" in r.stderr
        ref = open(tmp_path / ("ref_%s.tsv" % tag)).read()
        assert len(ref.splitlines()) > 300
        assert open(tmp_path / ("hip_%s.tsv" % tag)).read() == ref, tag
    # no parity flags at all (tantan + motif soft masking + --algo auto on both sides), every sensitivity
    for mode in ([], ["--fast"], ["--mid-sensitive"], ["--sensitive"], ["--more-sensitive"], ["--very-sensitive"], ["--ultra-sensitive"]):
        _run([REF, "blastp"] + mode + ["-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.faa"), "-o", str(tmp_path / "ref_plain.tsv"), "-p", "4"])
        _run([CLI, "blastp"] + mode + ["-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.faa"), "-o", str(tmp_path / "hip_plain.tsv"), "-p", "4"])
        assert open(tmp_path / "hip_plain.tsv").read() == open(tmp_path / "ref_plain.tsv").read(), mode
    # the reference's own choice of seed-index algorithm (AUTO picks the query-indexed path at these sizes) gives the same text
    _run([REF, "blastp", "--motif-masking", "0", "-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.faa"), "-o", str(tmp_path / "ref_auto.tsv"), "-p", "4"])
    assert open(tmp_path / "ref_auto.tsv").read() == open(tmp_path / "hip_default.tsv").read()
    # masking really changes this workload: the unmasked run differs
    _run([REF, "blastp", "--algo", "0", "--masking", "0", "--motif-masking", "0", "-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.faa"),
          "-o", str(tmp_path / "ref_nomask.tsv"), "-p", "4"])
    assert open(tmp_path / "ref_nomask.tsv").read() != open(tmp_path / "ref_default.tsv").read()


def test_cli_refuses_unimplemented_modes(tmp_path):
    r = subprocess.run([CLI, "blastp", "--faster", "-q", "x", "-d", "y"], capture_output=True, text=True)
    assert r.returncode != 0 and "not available" in r.stderr


def test_cli_blocked_matches_reference(tmp_path):
    """-b: several query and reference blocks (reference ctest diamond-test-blastp-blocked: -c1 -b0.00002), joined as
    join_blocks does. Same text as the reference with the same block size."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    rng = np.random.default_rng(33)
    db, doff, q, qoff = synth.generate(400, members=10, queries=500, seed=33)
    db, q = _plant_repeats(db, doff, rng), _plant_repeats(q, qoff, rng)
    synth.write_fasta(str(tmp_path / "db.faa"), "t", db, doff)
    synth.write_fasta(str(tmp_path / "q.faa"), "q", q, qoff)
    _run([CLI, "makedb", "--in", str(tmp_path / "db.faa"), "-d", str(tmp_path / "db")])
    common = ["--motif-masking", "0", "-q", str(tmp_path / "q.faa"), "-p", "4"]
    cases = [("fast_fasta", ["--fast", "-b0.0001", "-c1"], "db.faa"), ("default_dmnd", ["-b", "0.00015", "-c1"], "db.dmnd"),
             ("default_c4", ["-b0.0003"], "db.dmnd"), ("fast_k3", ["--fast", "-b0.0001", "-c1", "-k", "3"], "db.dmnd"), ("sensitive", ["--sensitive", "-b0.0002", "-c1"], "db.faa")]
    for tag, mode, dbfile in cases:
        _run([REF, "blastp", "--algo", "0"] + mode + common + ["-d", str(tmp_path / dbfile), "-o", str(tmp_path / ("ref_%s.tsv" % tag))])
        r = _run([CLI, "blastp"] + mode + common + ["-d", str(tmp_path / dbfile), "-o", str(tmp_path / ("hip_%s.tsv" % tag))])
        assert "reference blocks=" in r.stderr
        ref = open(tmp_path / ("ref_%s.tsv" % tag)).read()
        assert len(ref.splitlines()) > 300
        assert open(tmp_path / ("hip_%s.tsv" % tag)).read() == ref, tag
    assert max(np.unique([l.split("\t")[0] for l in open(tmp_path / "hip_fast_k3.tsv")], return_counts=True)[1]) == 3      # the join's cap
    _run([CLI, "blastp", "-c1"] + common + ["-d", str(tmp_path / "db.dmnd"), "-o", str(tmp_path / "hip_single.tsv")])
    _run([REF, "blastp", "--algo", "0", "-c1"] + common + ["-d", str(tmp_path / "db.dmnd"), "-o", str(tmp_path / "ref_single.tsv")])
    single = open(tmp_path / "hip_single.tsv").read()
    assert single == open(tmp_path / "ref_single.tsv").read()
    # (with <= 25 detectable targets per query the blocked and the single-block text coincide here; in general they need not)
    assert sorted(single.splitlines()) == sorted(open(tmp_path / "hip_default_dmnd.tsv").read().splitlines())
    # blastx: reads are cut into blocks by the letters of their surviving ORFs
    dna, off = synth.back_translate(q[:qoff[250]], qoff[:251], seed=12)
    synth.write_dna_fasta(str(tmp_path / "reads.fna"), "r", dna, off)
    xargs = ["-b0.0001", "-c1", "--masking", "0", "--motif-masking", "0", "-q", str(tmp_path / "reads.fna"), "-d", str(tmp_path / "db.faa"), "-p", "4"]
    _run([REF, "blastx", "--algo", "0"] + xargs + ["-o", str(tmp_path / "ref_x.tsv")])
    _run([CLI, "blastx"] + xargs + ["-o", str(tmp_path / "hip_x.tsv")])
    ref = open(tmp_path / "ref_x.tsv").read()
    assert len(ref.splitlines()) > 150
    assert open(tmp_path / "hip_x.tsv").read() == ref


def test_cli_query_indexed_matches_reference(tmp_path):
    """--algo 1 on sequences with masked runs (also at sequence starts), stop codons and ambiguity letters, where the hashed
    seeds of the query-indexed algorithm differ from the spaced seeds of --algo 0: three sensitivities without masking, and
    default masking (tantan, applied lazily to the targets by the reference)."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    g = os.path.join(ROOT, "tests", "golden")
    for q, db in (("hashed_q.faa", "hashed_db.faa"), ("hashed_sens_q.faa", "hashed_sens_db.faa")):
        for sens in (["--fast"], [], ["--sensitive"]):
            for masking in ("0", "tantan"):
                args = ["blastp"] + sens + ["--algo", "1", "--masking", masking, "-q", os.path.join(g, q), "-d", os.path.join(g, db), "-p", "4"]
                _run([REF] + args + ["--motif-masking", "0", "-o", str(tmp_path / "ref.tsv")])
                _run([CLI] + args + ["-o", str(tmp_path / "hip.tsv")])
                ref = open(tmp_path / "ref.tsv").read()
                assert len(ref.splitlines()) > 20
                assert open(tmp_path / "hip.tsv").read() == ref, (q, sens, masking)
    # the committed reference outputs of the golden runs (tests/golden/hashed*.tsv, minted with --masking 0)
    _run([CLI, "blastp", "--sensitive", "--algo", "1", "--masking", "0", "-q", os.path.join(g, "hashed_sens_q.faa"), "-d", os.path.join(g, "hashed_sens_db.faa"), "-o", str(tmp_path / "s.tsv"), "-p", "2"])
    assert open(tmp_path / "s.tsv").read() == open(os.path.join(g, "hashed_sens.tsv")).read()


# the reference's own ctest cases on its 389-domain fixture (CMakeLists.txt:553-572: `diamond ARGS -o NAME.out`, then a plain
# diff against src/test/NAME.out) that this build's options cover; fixtures copied to tests/golden/ref_ctest/
CTEST = [
    ("default", ["-p1"]),
    ("multithreaded", ["-p4"]),
    ("blocked", ["-c1", "-b0.00002", "-p4"]),
    ("more-sensitive", ["--more-sensitive", "-c1", "-p4"]),
    ("very-sensitive", ["--very-sensitive", "-c1", "-p4"]),
    ("ultra-sensitive", ["--ultra-sensitive", "-c1", "-p4"]),
    ("query-indexed", ["--more-sensitive", "-c1", "-p4", "--algo", "1"]),
    ("comp-based-stats-0", ["--more-sensitive", "-c1", "-p4", "--comp-based-stats", "0"]),
    ("comp-based-stats-2", ["--more-sensitive", "-c1", "-p4", "--comp-based-stats", "2"]),
    ("comp-based-stats-3", ["--more-sensitive", "-c1", "-p4", "--comp-based-stats", "3"]),
    ("comp-based-stats-4", ["--more-sensitive", "-c1", "-p4", "--comp-based-stats", "4"]),
    ("target-seqs", ["-k3", "-c1", "-p4"]),
    ("evalue", ["-e10000", "--more-sensitive", "-c1", "-p4"]),
    ("top", ["--top", "10", "-p4"]),
    ("pairwise-format", ["-c1", "-f0", "-p4"]),
    ("paf-format", ["-c1", "-f", "paf", "-p1"]),
]


@pytest.mark.parametrize("name,args", CTEST, ids=[c[0] for c in CTEST])
def test_cli_reproduces_reference_ctest_golden(tmp_path, name, args):
    """DEFAULT command lines: tantan masking, motif soft masking, --algo auto -- no parity flags on either side."""
    g = os.path.join(ROOT, "tests", "golden", "ref_ctest")
    assert os.path.exists(os.path.join(ROOT, "diamond_amd", "motifs.bin")), "motif table not generated (tools/make_motif_table.py)"
    out = str(tmp_path / "out")
    _run([CLI, "blastp", "-q", os.path.join(g, "data.faa"), "-d", os.path.join(g, "data.faa")] + args + ["-o", out])
    path = os.path.join(g, "diamond-test-blastp-%s.out" % name)
    if os.path.exists(path + ".gz"):
        import gzip
        want = gzip.open(path + ".gz", "rt").read()
    else:
        want = open(path).read()
    got = open(out).read()
    if got != want:
        a, b = set(want.splitlines()), set(got.splitlines())
        raise AssertionError("%s: %d lines only in the golden, %d only in ours; e.g. %s | %s" % (name, len(a - b), len(b - a), sorted(a - b)[:3], sorted(b - a)[:3]))


def test_cli_motif_masking_matches_reference_on_planted_motifs(tmp_path):
    """Synthetic sequences with motifs of the reference's table planted in queries and targets (single, overlapping, chained
    beyond max_motif_len, covering more than half of a short sequence): default flags on both sides, three sensitivities,
    both algorithms."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    import struct
    raw = open(os.path.join(ROOT, "diamond_amd", "motifs.bin"), "rb").read()
    codes = struct.unpack("<%dQ" % (len(raw) // 8), raw)
    rng = np.random.default_rng(12)

    def letters(code):
        out = []
        for _ in range(8):
            out.append(code % 20)
            code //= 20
        return np.array(out[::-1], np.int8)

    db, doff, q, qoff = synth.generate(300, members=8, queries=400, seed=21)

    def plant(data, off):
        data = data.copy()
        for i in range(len(off) - 1):
            b, e = int(off[i]), int(off[i + 1])
            r = rng.random()
            if r < 0.5:
                for _ in range(int(rng.integers(1, 4))):
                    p = int(rng.integers(b, max(b + 1, e - 8)))
                    data[p:p + 8] = letters(codes[int(rng.integers(0, len(codes)))])[:max(0, min(8, e - p))]
            elif r < 0.6 and e - b > 60:                        # 5 motifs back to back: a 40-letter range, too long to mask
                p = int(rng.integers(b, e - 41))
                for k in range(5):
                    data[p + 8 * k:p + 8 * k + 8] = letters(codes[int(rng.integers(0, len(codes)))])
        return data

    synth.write_fasta(str(tmp_path / "db.faa"), "t", plant(db, doff), doff)
    synth.write_fasta(str(tmp_path / "q.faa"), "q", plant(q, qoff), qoff)
    for sens in (["--fast"], [], ["--sensitive"]):
        for algo in ("0", "1"):
            args = ["blastp"] + sens + ["--algo", algo, "-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.faa"), "-p", "4"]
            _run([REF] + args + ["-o", str(tmp_path / "ref.tsv")])
            log = _run([CLI] + args + ["-o", str(tmp_path / "hip.tsv")])
            ref = open(tmp_path / "ref.tsv").read()
            assert len(ref.splitlines()) > 300
            assert open(tmp_path / "hip.tsv").read() == ref, (sens, algo)
    assert "Soft-masked letters (motifs):" in log.stderr


@pytest.mark.parametrize("flags,golden", [(["--algo", "0"], "motif.tsv"), (["--fast", "--algo", "1"], "motif_a1.tsv")])
def test_cli_reproduces_motif_masking_golden(tmp_path, flags, golden):
    """The committed output of the genuine reference with its default flags on sequences with planted motifs
    (tests/golden/make_motif_golden.sh)."""
    g = os.path.join(ROOT, "tests", "golden")
    out = str(tmp_path / "hip.tsv")
    log = _run([CLI, "blastp"] + flags + ["-q", os.path.join(g, "motif_q.faa"), "-d", os.path.join(g, "motif_db.faa"), "-o", out, "-p", "2"])
    assert "Soft-masked letters (motifs):" in log.stderr
    assert open(out).read() == open(os.path.join(g, golden)).read()


ALL_FIELDS = ["qseqid", "qlen", "sseqid", "sallseqid", "slen", "qstart", "qend", "sstart", "send", "qseq", "sseq", "evalue", "bitscore", "score",
              "length", "pident", "nident", "mismatch", "positive", "gapopen", "gaps", "ppos", "qframe", "btop", "stitle", "salltitles", "qcovhsp",
              "qtitle", "full_sseq", "qnum", "snum", "scovhsp", "full_qseq", "qseq_gapped", "sseq_gapped", "qstrand", "cigar"]


@pytest.mark.parametrize("mode", ["blastp", "blastp-sensitive-blocked", "blastx"])
def test_cli_output_fields_and_pairwise_match_reference(tmp_path, mode):
    """`-f 6 FIELD...` with every field this build prints, and `-f 0`, against the reference binary on the same files (default
    flags: tantan-masked letters show up as X in qseq / sseq, full_sseq prints the unmasked target)."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    db, doff, q, qoff = synth.generate(150, members=6, queries=200, seed=31)
    rng = np.random.default_rng(4)
    for a, off in ((db, doff), (q, qoff)):                       # low-complexity stretches that tantan masks
        for i in range(0, len(off) - 1, 7):
            b, e = int(off[i]), int(off[i + 1])
            if e - b > 80:
                p = int(rng.integers(b, e - 40))
                a[p:p + 36] = np.tile(np.array([int(rng.integers(0, 20)), int(rng.integers(0, 20))], np.int8), 18)
    synth.write_fasta(str(tmp_path / "db.faa"), "t", db, doff)
    titles = str(tmp_path / "db2.faa")                            # titles with a description and a second \1-separated id
    with open(tmp_path / "db.faa") as f, open(titles, "w") as g:
        for n, line in enumerate(f):
            g.write(line.rstrip("\n") + (" some protein [organism %d]\x01alt%d second title\n" % (n, n) if line.startswith(">") else "\n"))
    if mode == "blastx":
        dna, dna_off = synth.back_translate(q[:qoff[120]], qoff[:121], seed=5)
        synth.write_dna_fasta(str(tmp_path / "q.fa"), "r", dna, dna_off)
        base = ["blastx", "-q", str(tmp_path / "q.fa"), "-d", titles, "-p", "4"]
        fields = ALL_FIELDS + ["qseq_translated"]
    else:
        synth.write_fasta(str(tmp_path / "q.faa"), "q", q, qoff)
        base = ["blastp", "-q", str(tmp_path / "q.faa"), "-d", titles, "-p", "4"]
        if mode == "blastp-sensitive-blocked":
            base += ["--sensitive", "-b0.00003", "-c1"]
        fields = ALL_FIELDS
    for name, fmt in (("fields", ["-f", "6"] + fields), ("pairwise", ["-f", "0"]), ("paf", ["-f", "paf"]), ("sam", ["-f", "sam"])):
        _run([REF] + base + fmt + ["-o", str(tmp_path / ("ref_" + name))])
        _run([CLI] + base + fmt + ["-o", str(tmp_path / ("hip_" + name))])
        want, got = open(tmp_path / ("ref_" + name)).read(), open(tmp_path / ("hip_" + name)).read()
        if name == "sam":                                        # the header names the program, its version and command line
            assert got.startswith("@HD\tVN:1.5\tSO:query\n@PG\tPN:diamond-hip")
            want, got = ("\n".join(l for l in t.splitlines() if not l.startswith("@")) for t in (want, got))
        assert len(want.splitlines()) > 200
        if got != want:
            w, g2 = want.splitlines(), got.splitlines()
            k = next((i for i in range(min(len(w), len(g2))) if w[i] != g2[i]), min(len(w), len(g2)))
            if name == "fields" and k < min(len(w), len(g2)):
                a, b = w[k].split("\t"), g2[k].split("\t")
                bad = [(fields[i], a[i][:60], b[i][:60]) for i in range(min(len(a), len(b))) if a[i] != b[i]]
                raise AssertionError("%s %s line %d: %s" % (mode, name, k, bad[:4]))
            raise AssertionError("%s %s: first difference at line %d of %d/%d:\n%r\n%r" % (mode, name, k, len(w), len(g2), w[k:k + 2], g2[k:k + 2]))
    assert "X" in "".join(l.split("\t")[9] for l in open(tmp_path / "ref_fields")) or mode == "blastx"


def test_cli_gpus_distributes_reference_blocks(tmp_path):
    """`--gpus 2`: the reference blocks go to two host threads with a context each (here both on the one GPU of the box,
    DMND_CLI_SHARE_GPU=1) and the records are joined as for a -b run. (a) with the -b of the reference's own `blocked` ctest the
    output is that golden; (b) without -b the database is cut into two blocks: same text as the explicit cut on one GPU, and as
    the reference run with that -b."""
    g = os.path.join(ROOT, "tests", "golden", "ref_ctest")
    env = dict(os.environ, DMND_CLI_SHARE_GPU="1")
    out = str(tmp_path / "out")
    r = subprocess.run([CLI, "blastp", "-q", os.path.join(g, "data.faa"), "-d", os.path.join(g, "data.faa"), "-c1", "-b0.00002", "-p4", "--gpus", "2", "-o", out],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(out).read() == open(os.path.join(g, "diamond-test-blastp-blocked.out")).read()
    # round 5: with several contexts the records are merged by dmnd_join_ranks -- owners by query range, exchange, merge on the device.
    # Two contexts on the box's one GPU exchange with device copies (RCCL refuses two ranks on a device) ...
    assert "Block join: transport_used = device-to-device copies between 2 context(s), merged on the device(s)" in r.stderr, r.stderr[-1500:]
    # ... and ONE context goes through RCCL itself (ncclCommInitAll, a grouped ncclSend / ncclRecv to itself): same golden; so does --top
    for extra, golden in (([], "diamond-test-blastp-blocked.out"), ):
        r1 = subprocess.run([CLI, "blastp", "-q", os.path.join(g, "data.faa"), "-d", os.path.join(g, "data.faa"), "-c1", "-b0.00002", "-p4", "-o", out] + extra,
                            capture_output=True, text=True, timeout=600, env=dict(os.environ, DMND_CLI_RCCL="1"))
        assert r1.returncode == 0, r1.stderr[-2000:]
        assert "Block join: transport_used = RCCL exchange (ncclSend/ncclRecv) between 1 context(s)" in r1.stderr, r1.stderr[-1500:]
        assert open(out).read() == open(os.path.join(g, golden)).read()
    for top_flags in (["--top", "10"], ["-k", "3"]):
        outs = []
        for e in (dict(DMND_CLI_RCCL="1"), dict(DMND_CLI_RCCL="0"), dict(DMND_CLI_SHARE_GPU="1", GPUS="2")):
            cmd = [CLI, "blastp", "-q", os.path.join(g, "data.faa"), "-d", os.path.join(g, "data.faa"), "-c1", "-b0.00002", "-p4", "-o", out] + top_flags + (["--gpus", e.pop("GPUS")] if "GPUS" in e else [])
            rr = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, **e))
            assert rr.returncode == 0, rr.stderr[-2000:]
            outs.append(open(out).read())
        assert outs[0] == outs[1] == outs[2] and len(outs[0]) > 1000, top_flags
    for fmt in ([], ["-f", "6", "qseqid", "sseqid", "evalue", "cigar", "btop"]):
        r = subprocess.run([CLI, "blastp", "-q", os.path.join(g, "data.faa"), "-d", os.path.join(g, "data.faa"), "-p4", "--gpus", "2", "-o", out] + fmt,
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "reference blocks=2" in r.stderr
        letters = sum(len(l.strip()) for l in open(os.path.join(g, "data.faa")) if not l.startswith(">"))
        b = "%.12f" % (((letters + 1) // 2 + 0.5) / 1e9)
        assert int(float(b) * 1e9) == (letters + 1) // 2
        one = str(tmp_path / "one")
        _run([CLI, "blastp", "-q", os.path.join(g, "data.faa"), "-d", os.path.join(g, "data.faa"), "-p4", "-b", b, "-o", one] + fmt)
        assert open(out).read() == open(one).read()
        if os.path.exists(REF):
            ref = str(tmp_path / "ref")
            _run([REF, "blastp", "-q", os.path.join(g, "data.faa"), "-d", os.path.join(g, "data.faa"), "-p4", "-b", b, "-o", ref] + fmt)
            assert open(ref).read() == open(out).read()
    r = subprocess.run([CLI, "blastp", "-q", os.path.join(g, "data.faa"), "-d", os.path.join(g, "data.faa"), "--gpus", "64", "-o", out], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "gfx950 device(s) visible" in r.stderr


def _write_repeat_protein_files(tmp_path, indel=0.02):
    """Families plus long tandem-repeat proteins and multi-domain proteins with long unrelated insertions (db.faa, q.faa)."""
    rng = np.random.default_rng(8)
    db, doff, q, qoff = synth.generate(40, members=4, queries=40, seed=9, indel=indel)
    seqs_t = [db[doff[i]:doff[i + 1]] for i in range(len(doff) - 1)]
    seqs_q = [q[qoff[i]:qoff[i + 1]] for i in range(len(qoff) - 1)]

    def repeat_protein(unit, n, rate):
        s = np.tile(unit, n).copy()
        mut = rng.random(len(s)) < rate
        s[mut] = rng.integers(0, 20, int(mut.sum()))
        return s

    for k in range(3):
        unit = rng.integers(0, 20, 61 + 10 * k).astype(np.int8)
        seqs_t.append(repeat_protein(unit, 90, 0.15))
        seqs_t.append(repeat_protein(unit, 70, 0.25))
        seqs_q.append(repeat_protein(unit, 80, 0.2))
    # domains in the same order with long unrelated insertions between them in the target: one chain that drifts over
    # thousands of diagonals
    for k in range(3):
        doms = [rng.integers(0, 20, 350).astype(np.int8) for _ in range(5)]
        seqs_q.append(np.concatenate(doms))
        parts = []
        for d in doms:
            m = d.copy()
            mut = rng.random(len(m)) < 0.1
            m[mut] = rng.integers(0, 20, int(mut.sum()))
            parts += [m, rng.integers(0, 20, 1100 + 150 * k).astype(np.int8)]
        seqs_t.append(np.concatenate(parts[:-1]))
    for name, seqs, prefix in (("db.faa", seqs_t, "t"), ("q.faa", seqs_q, "q")):
        off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])])
        synth.write_fasta(str(tmp_path / name), prefix, np.concatenate(seqs), off)


def test_cli_long_repeat_proteins_with_wide_bands(tmp_path):
    """Long tandem-repeat proteins: the chains of a pair spread over thousands of diagonals and add_dp_targets merges them into
    bands wider than one wavefront sweeps (round 1 aborted the block pair with DMND_E_BAND). Whole pipeline against the reference."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    _write_repeat_protein_files(tmp_path)
    traces = []
    for sens in (["--fast"], ["--sensitive"]):
        args = ["blastp"] + sens + ["-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.faa"), "-p", "4", "--masking", "0"]
        _run([REF] + args + ["-o", str(tmp_path / "ref.tsv")])
        r = subprocess.run([CLI] + args + ["-o", str(tmp_path / "hip.tsv")], capture_output=True, text=True, timeout=600, env=dict(os.environ, DMND_TRACE="1"))
        assert r.returncode == 0, r.stderr[-2000:]
        traces.append(r.stderr)
        ref = open(tmp_path / "ref.tsv").read()
        assert sum(1 for l in ref.splitlines() if int(l.split("\t")[3]) > 3000) >= 3      # the long repeat alignments are reported
        assert open(tmp_path / "hip.tsv").read() == ref, sens
    print("\n".join(l for t in traces for l in t.splitlines() if "wavefronts each" in l or "band" in l.lower()))
    assert any("wavefronts each" in t for t in traces)        # some band was wider than one wavefront sweeps
    # output that needs the transcript: matrices above max_swipe_dp cells are traced too (the statistics cells only replace the
    # traceback when no transcript is asked for, swipe_wrapper.cpp:91-96)
    for fmt in (["-f", "6", "qseqid", "sseqid", "length", "gapopen", "gaps", "btop", "cigar"], ["-f", "0"]):
        args = ["blastp", "-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.faa"), "-p", "4", "--masking", "0"] + fmt
        _run([REF] + args + ["-o", str(tmp_path / "ref.out")])
        _run([CLI] + args + ["-o", str(tmp_path / "hip.out")])
        ref = open(tmp_path / "ref.out").read()
        assert len(ref) > 50000 and open(tmp_path / "hip.out").read() == ref, fmt


def test_cli_top_percent_and_large_k_match_reference(tmp_path):
    """--top N (bit-score window instead of -k, also over several reference blocks) and -k above MAX_CHUNK_SIZE (the first ranking
    chunk grows by the seed-hit e-value rule, extend.cpp:262-268) against the reference binary."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    db, doff, q, qoff = synth.generate(60, members=30, queries=120, seed=17)          # large families: many targets per query
    synth.write_fasta(str(tmp_path / "db.faa"), "t", db, doff)
    synth.write_fasta(str(tmp_path / "q.faa"), "q", q, qoff)
    base = ["blastp", "-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.faa"), "-p", "4"]
    for extra in (["--top", "5"], ["--top", "30", "--fast"], ["--top", "50", "-b0.0002", "-c1"], ["-k", "500", "--sensitive"], ["-k", "450", "-e", "10"]):
        _run([REF] + base + extra + ["-o", str(tmp_path / "ref.tsv")])
        _run([CLI] + base + extra + ["-o", str(tmp_path / "hip.tsv")])
        ref = open(tmp_path / "ref.tsv").read()
        assert len(ref.splitlines()) > 100, extra
        assert open(tmp_path / "hip.tsv").read() == ref, extra


def test_cli_identity_cover_and_min_score_filters_match_reference(tmp_path):
    """--id / --query-cover / --subject-cover (HSPs removed after round 2; the extension then takes more of the ranked targets, a
    step at a time, until -k matches pass) and --min-score (bit-score report cutoff) against the reference binary."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    db, doff, q, qoff = synth.generate(60, members=40, queries=150, seed=23)
    rng = np.random.default_rng(3)
    seqs = [db[doff[i]:doff[i + 1]] for i in range(len(doff) - 1)]
    for i in range(0, len(seqs), 3):                                     # truncated members: low query / subject cover
        cut = int(rng.integers(len(seqs[i]) // 3, len(seqs[i])))
        seqs[i] = seqs[i][:cut] if i % 2 else seqs[i][len(seqs[i]) - cut:]
    off = np.concatenate([[0], np.cumsum([len(x) for x in seqs])])
    synth.write_fasta(str(tmp_path / "db.faa"), "t", np.concatenate(seqs), off)
    synth.write_fasta(str(tmp_path / "q.faa"), "q", q, qoff)
    base = ["blastp", "-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.faa"), "-p", "4"]
    plain = None
    for extra in ([], ["--id", "55"], ["--query-cover", "80", "-k", "5"], ["--subject-cover", "90", "-k", "3", "--fast"], ["--id", "50", "--query-cover", "60", "--subject-cover", "65", "-k", "40"],
                  ["--min-score", "150"], ["--id", "60", "--top", "20"], ["--query-cover", "75", "-b0.0003", "-c1", "--sensitive", "-k", "10"]):
        _run([REF] + base + extra + ["-o", str(tmp_path / "ref.tsv")])
        _run([CLI] + base + extra + ["-o", str(tmp_path / "hip.tsv")])
        ref = open(tmp_path / "ref.tsv").read()
        assert len(ref.splitlines()) > 100, extra
        got = open(tmp_path / "hip.tsv").read()
        if got != ref:
            a, b = ref.splitlines(), got.splitlines()
            k = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), min(len(a), len(b)))
            raise AssertionError("%s: %d vs %d lines, first difference at %d: %r | %r; only ref %d, only ours %d" % (extra, len(a), len(b), k, a[k:k + 2], b[k:k + 2], len(set(a) - set(b)), len(set(b) - set(a))))
        if not extra:
            plain = ref
        else:
            assert ref != plain, extra                                    # the option matters on this input


def test_cli_unaligned_queries_header_and_translated_query_cover(tmp_path):
    """--unal 1 (tabular lines of queries without alignments: those that had seed hits with one reference block, all of them
    with several), --unal 0 for the formats that report them by default, --header simple, and --query-cover for blastx (measured
    on the DNA read) against the reference binary."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    rng = np.random.default_rng(41)
    db, doff, q, qoff = synth.generate(120, members=6, queries=150, seed=41)
    seqs = [q[qoff[i]:qoff[i + 1]] for i in range(len(qoff) - 1)]
    for i in range(0, len(seqs), 4):                                       # unrelated queries: most get no alignment
        seqs[i] = rng.integers(0, 20, len(seqs[i])).astype(np.int8)
    off = np.concatenate([[0], np.cumsum([len(x) for x in seqs])])
    qq = np.concatenate(seqs)
    synth.write_fasta(str(tmp_path / "db.faa"), "t", db, doff)
    synth.write_fasta(str(tmp_path / "q.faa"), "q", qq, off)
    dna, dna_off = synth.back_translate(qq[:off[100]], off[:101], seed=5)
    synth.write_dna_fasta(str(tmp_path / "q.fa"), "r", dna, dna_off)
    p = ["-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.faa"), "-p", "4"]
    x = ["-q", str(tmp_path / "q.fa"), "-d", str(tmp_path / "db.faa"), "-p", "4"]
    cases = [(["blastp"] + p + ["--unal", "1"], 150), (["blastp"] + p + ["--unal", "1", "-b0.00002", "-c1"], 150),
             (["blastp"] + p + ["--unal", "1", "-f", "6", "qseqid", "qlen", "sseqid", "evalue", "qtitle", "full_qseq", "cigar", "qframe", "slen"], 150),
             (["blastp"] + p + ["--unal", "0", "-f", "0"], 100), (["blastp"] + p + ["--unal", "0", "-f", "sam"], 100), (["blastp"] + p + ["--header", "simple", "-f", "6", "qseqid", "sseqid", "bitscore"], 100),
             (["blastx"] + x + ["--unal", "1", "-f", "6", "qseqid", "qlen", "sseqid", "qstart", "qend", "full_qseq"], 60),
             (["blastx"] + x + ["--query-cover", "70"], 20), (["blastx"] + x + ["--query-cover", "85", "--id", "60", "-k", "3"], 5)]
    for args, min_lines in cases:
        _run([REF] + args + ["-o", str(tmp_path / "ref.out")])
        _run([CLI] + args + ["-o", str(tmp_path / "hip.out")])
        want, got = open(tmp_path / "ref.out").read(), open(tmp_path / "hip.out").read()
        if "sam" in args:
            want, got = ("\n".join(l for l in t.splitlines() if not l.startswith("@")) for t in (want, got))
        assert len(want.splitlines()) >= min_lines, args
        if got != want:
            a, b = want.splitlines(), got.splitlines()
            k = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), min(len(a), len(b)))
            raise AssertionError("%s: %d vs %d lines, first difference at %d: %r | %r" % (args[5:], len(a), len(b), k, a[k:k + 1], b[k:k + 1]))


def test_cli_runs_without_any_hit_and_with_tiny_inputs(tmp_path):
    """No seed hit at all, a query shorter than a seed, one sequence on each side: every output format finishes with the reference's
    text (possibly empty)."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    rng = np.random.default_rng(9)
    with open(tmp_path / "q.faa", "w") as f:
        f.write(">a first\n" + "".join("ARNDCQEGHILKMFPSTWYV"[int(x)] for x in rng.integers(0, 20, 90)) + "\n>b\nMKV\n>c\n" + "A" * 40 + "\n")
    with open(tmp_path / "db.faa", "w") as f:
        f.write(">t0 only target\n" + "".join("ARNDCQEGHILKMFPSTWYV"[int(x)] for x in rng.integers(0, 20, 120)) + "\n")
    base = ["blastp", "-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.faa"), "-p", "2"]
    for extra in ([], ["--unal", "1"], ["-f", "0"], ["-f", "paf"], ["-f", "sam"], ["--sensitive", "-f", "6", "qseqid", "sseqid", "cigar"], ["--top", "10"], ["--id", "30", "--unal", "1"],
                  ["-b0.00000005", "--unal", "1"]):
        _run([REF] + base + extra + ["-o", str(tmp_path / "ref.out")])
        _run([CLI] + base + extra + ["-o", str(tmp_path / "hip.out")])
        want, got = open(tmp_path / "ref.out").read(), open(tmp_path / "hip.out").read()
        if "sam" in extra:
            want, got = ("\n".join(l for l in t.splitlines() if not l.startswith("@")) for t in (want, got))
        assert got == want, (extra, want[:300], got[:300])
    # a self hit exists when the target is among the queries
    with open(tmp_path / "db2.faa", "w") as f:
        f.write(open(tmp_path / "q.faa").read())
    args = ["blastp", "-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db2.faa"), "-p", "2", "--unal", "1"]
    _run([REF] + args + ["-o", str(tmp_path / "ref.out")])
    _run([CLI] + args + ["-o", str(tmp_path / "hip.out")])
    assert open(tmp_path / "hip.out").read() == open(tmp_path / "ref.out").read() != ""


def test_cli_no_self_hits_matches_reference(tmp_path):
    """--no-self-hits on an all-against-all search of the reference's own fixture (every query is in the database under its own
    title), also with several reference blocks and another output format; duplicates under a different title stay."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    g = os.path.join(ROOT, "tests", "golden", "ref_ctest", "data.faa")
    lines = open(g).read().splitlines()
    with open(tmp_path / "db.faa", "w") as f:                       # + copies of the first sequences under other titles
        f.write("\n".join(lines) + "\n")
        k = 0
        for i, l in enumerate(lines[:40]):
            if l.startswith(">"):
                f.write(">copy%d of %s\n%s\n" % (k, l[1:], lines[i + 1]))
                k += 1
    base = ["blastp", "-q", g, "-d", str(tmp_path / "db.faa"), "-p", "4", "--no-self-hits"]
    plain = None
    for extra in ([], ["-k", "3"], ["-b0.00002", "-c1"], ["-f", "6", "qseqid", "sseqid", "pident", "stitle"], ["--id", "40"]):
        _run([REF] + base + extra + ["-o", str(tmp_path / "ref.tsv")])
        _run([CLI] + base + extra + ["-o", str(tmp_path / "hip.tsv")])
        ref = open(tmp_path / "ref.tsv").read()
        assert len(ref.splitlines()) > 30, extra
        assert open(tmp_path / "hip.tsv").read() == ref, extra
        if not extra:
            plain = ref
    assert not any(l.split("\t")[0] == l.split("\t")[1] for l in plain.splitlines())       # no self hit
    assert any(l.split("\t")[1].startswith("copy") and l.split("\t")[2] == "100" for l in plain.splitlines())      # the renamed copies are reported


def test_cli_scoring_matrices_and_gap_penalties_match_reference(tmp_path):
    """--matrix (the eight standard matrices) / --gapopen / --gapextend: default flags (tantan + motif masking use the matrix too) on
    the reference's ctest fixture, the filters of --sensitive and a translated search on synthetic families, the positives of the
    output formats."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    g = os.path.join(ROOT, "tests", "golden", "ref_ctest", "data.faa")
    db, doff, q, qoff = synth.generate(300, members=10, queries=300, seed=23)
    dna, off = synth.back_translate(q[: qoff[120]], qoff[:121], seed=24)
    synth.write_fasta(str(tmp_path / "db.faa"), "t", db, doff)
    synth.write_fasta(str(tmp_path / "q.faa"), "q", q, qoff)
    synth.write_dna_fasta(str(tmp_path / "reads.fna"), "r", dna, off)
    fixture = ["blastp", "-q", g, "-d", g, "-p", "4"]
    fam = ["blastp", "-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.faa"), "-p", "4"]
    runs = [(fixture + ["--matrix", m], 5 if m == "pam250" else 500) for m in ("blosum45", "BLOSUM50", "blosum80", "blosum90", "pam30", "PAM70", "pam250")]
    runs += [(fixture + ["--gapopen", "9", "--gapextend", "2"], 500), (fixture + ["--matrix", "blosum45", "--gapopen", "19", "--gapextend", "1", "--masking", "0"], 500),
             (fam + ["--matrix", "pam30", "--sensitive"], 100), (fam + ["--matrix", "blosum45", "--sensitive"], 300),
             (fam + ["--matrix", "blosum90", "--fast", "--comp-based-stats", "0"], 100), (fam + ["--gapopen", "6", "--gapextend", "2", "--very-sensitive"], 300),
             (fam + ["--matrix", "blosum50", "-f", "6", "qseqid", "sseqid", "positive", "ppos", "score", "bitscore", "evalue", "cigar"], 300),
             (fam + ["--matrix", "pam70", "-f", "0"], 1000),
             (["blastx", "-q", str(tmp_path / "reads.fna"), "-d", str(tmp_path / "db.faa"), "-p", "4", "--matrix", "blosum80"], 100)]
    bad = []
    for args, least in runs:
        _run([REF] + args + ["-o", str(tmp_path / "ref.out")])
        _run([CLI] + args + ["-o", str(tmp_path / "hip.out")])
        ref, hip_ = open(tmp_path / "ref.out").read(), open(tmp_path / "hip.out").read()
        if len(ref.splitlines()) < least or ref != hip_:
            r, h = ref.splitlines(), hip_.splitlines()
            first = next((i for i in range(min(len(r), len(h))) if r[i] != h[i]), min(len(r), len(h)))
            bad.append("%s: ref %d lines, hip %d lines, first difference at %d\n  ref: %s\n  hip: %s" % (
                " ".join(args[5:]), len(r), len(h), first, r[first] if first < len(r) else "-", h[first] if first < len(h) else "-"))
    assert not bad, "\n".join(bad)
    for args, msg in ((fixture + ["--matrix", "blosum61"], "Unknown scoring matrix"), (fixture + ["--gapopen", "3"], "outside the supported range")):
        r = subprocess.run([CLI] + args + ["-o", str(tmp_path / "x.out")], capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and msg in r.stderr


def test_cli_input_formats_and_translation_options_match_reference(tmp_path):
    """gzip-compressed FASTA and FASTQ inputs (query, database, makedb), and the options of a translated search: --strand,
    --query-gencode, --min-orf. (A gzip-compressed FASTQ is read here too; the reference build loads no query from one.)"""
    import gzip
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    db, doff, q, qoff = synth.generate(300, members=10, queries=200, seed=31)
    dna, off = synth.back_translate(q[: qoff[150]], qoff[:151], seed=32)
    synth.write_fasta(str(tmp_path / "db.faa"), "t", db, doff)
    synth.write_fasta(str(tmp_path / "q.faa"), "q", q, qoff)
    synth.write_dna_fasta(str(tmp_path / "reads.fna"), "r", dna, off)

    def fastq(src, dst):
        recs = open(src).read().split(">")[1:]
        with open(dst, "w") as f:
            for r in recs:
                h, *s = r.strip().split("\n")
                s = "".join(s)
                f.write("@%s\n%s\n+%s\n%s\n" % (h, s, h if len(s) % 2 else "", "F" * len(s)))

    def gz(src):
        with open(src, "rb") as f, gzip.open(src + ".gz", "wb") as g:
            g.write(f.read())
        return src + ".gz"

    fastq(tmp_path / "q.faa", tmp_path / "q.fastq")
    fastq(tmp_path / "reads.fna", tmp_path / "reads.fastq")
    d = str(tmp_path / "db.faa")
    _run([REF, "blastp", "-q", str(tmp_path / "q.faa"), "-d", d, "-p", "4", "-o", str(tmp_path / "ref_p.tsv")])
    _run([REF, "blastx", "-q", str(tmp_path / "reads.fna"), "-d", d, "-p", "4", "-o", str(tmp_path / "ref_x.tsv")])
    ref_p, ref_x = open(tmp_path / "ref_p.tsv").read(), open(tmp_path / "ref_x.tsv").read()
    assert len(ref_p.splitlines()) > 300 and len(ref_x.splitlines()) > 100
    _run([CLI, "makedb", "--in", gz(d), "-d", str(tmp_path / "dbz")])
    for mode, query, database, want in (("blastp", gz(str(tmp_path / "q.faa")), d, ref_p), ("blastp", str(tmp_path / "q.fastq"), gz(d), ref_p),
                                        ("blastp", gz(str(tmp_path / "q.fastq")), str(tmp_path / "dbz.dmnd"), ref_p),
                                        ("blastx", gz(str(tmp_path / "reads.fna")), d, ref_x), ("blastx", str(tmp_path / "reads.fastq"), str(tmp_path / "dbz.dmnd"), ref_x)):
        _run([CLI, mode, "-q", query, "-d", database, "-p", "4", "-o", str(tmp_path / "hip.tsv")])
        assert open(tmp_path / "hip.tsv").read() == want, (mode, query, database)
    seen = set()
    for extra in (["--strand", "plus"], ["--strand", "minus"], ["--query-gencode", "4"], ["--min-orf", "10"], ["--query-gencode", "11", "--strand", "minus", "-l", "60"],
                  ["--query-gencode", "2", "--sensitive"]):
        args = ["blastx", "-q", str(tmp_path / "reads.fastq"), "-d", d, "-p", "4"] + extra
        _run([REF] + args + ["-o", str(tmp_path / "ref.tsv")])
        _run([CLI] + args + ["-o", str(tmp_path / "hip.tsv")])
        ref = open(tmp_path / "ref.tsv").read()
        assert len(ref.splitlines()) > 50, extra
        assert open(tmp_path / "hip.tsv").read() == ref, extra
        seen.add(ref)
    assert len(seen) >= 5 and ref_x not in seen                        # the options change the result
    r = subprocess.run([CLI, "blastx", "-q", str(tmp_path / "reads.fna"), "-d", d, "--query-gencode", "7", "-o", str(tmp_path / "x.tsv")], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "Invalid genetic code id" in r.stderr


def test_cli_unaligned_aligned_files_compressed_output_and_shape_count(tmp_path):
    """--un / --al (FASTA of the queries without / with alignments, masked letters as the block holds them, DNA reads for blastx),
    --compress 1 (gzip, ".gz" appended) and --shapes N against the reference."""
    import gzip
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    db, doff, q, qoff = synth.generate(200, members=10, queries=300, seed=41, decoy_frac=0.3)
    rng = np.random.default_rng(5)
    q = _plant_repeats(q, qoff, rng)                  # tantan masks these stretches: the --un / --al records show them as X
    dna, off = synth.back_translate(q[: qoff[120]], qoff[:121], seed=42)
    synth.write_fasta(str(tmp_path / "db.faa"), "t", db, doff)
    synth.write_fasta(str(tmp_path / "q.faa"), "q", q, qoff)
    synth.write_dna_fasta(str(tmp_path / "reads.fna"), "r", dna, off)
    for mode, query, extra in (("blastp", "q.faa", []), ("blastp", "q.faa", ["-b0.00003", "--id", "50"]), ("blastx", "reads.fna", ["--fast"])):
        outs = {}
        for tag, exe in (("ref", REF), ("hip", CLI)):
            _run([exe, mode, "-q", str(tmp_path / query), "-d", str(tmp_path / "db.faa"), "-p", "4", "-o", str(tmp_path / (tag + ".tsv")),
                  "--un", str(tmp_path / (tag + ".un")), "--al", str(tmp_path / (tag + ".al"))] + extra)
            outs[tag] = [open(tmp_path / (tag + e)).read() for e in (".tsv", ".un", ".al")]
        assert outs["hip"] == outs["ref"], (mode, extra)
        assert all(len(t) > 1000 for t in outs["ref"]), (mode, extra)
        if mode == "blastp":
            assert sum(l.count("XXXX") for l in outs["ref"][1].splitlines() + outs["ref"][2].splitlines() if not l.startswith(">")) > 5
    args = ["blastp", "-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.faa"), "-p", "4", "--compress", "1"]
    _run([REF] + args + ["-o", str(tmp_path / "refz.tsv")])
    _run([CLI] + args + ["-o", str(tmp_path / "hipz.tsv")])
    _run([CLI] + args + ["-o", str(tmp_path / "hipz2.tsv.gz"), "-f", "0"])
    ref = gzip.open(str(tmp_path / "refz.tsv.gz"), "rt").read()
    assert gzip.open(str(tmp_path / "hipz.tsv.gz"), "rt").read() == ref and len(ref) > 1000
    assert gzip.open(str(tmp_path / "hipz2.tsv.gz"), "rt").read().startswith("BLASTP 2.3.0+")
    for sens, n in (("--sensitive", "4"), ("--very-sensitive", "1"), ("--mid-sensitive", "40"), ("--fast", "1")):
        args = ["blastp", sens, "-s", n, "-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.faa"), "-p", "4"]
        _run([REF] + args + ["-o", str(tmp_path / "refs.tsv")])
        _run([CLI] + args + ["-o", str(tmp_path / "hips.tsv")])
        ref = open(tmp_path / "refs.tsv").read()
        assert len(ref.splitlines()) > 200 and open(tmp_path / "hips.tsv").read() == ref, (sens, n)


def test_cli_extension_modes_match_reference(tmp_path):
    """--ext banded-fast / banded-slow / full (whole-matrix alignment of every target with a seed hit, no chaining) on families with
    many indels, tandem-repeat and multi-domain proteins -- data on which the three modes give different alignments."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    _write_repeat_protein_files(tmp_path, indel=0.08)
    seen = {}
    for extra in (["--ext", "banded-fast"], ["--ext", "banded-slow"], ["--ext", "full"], ["--ext", "full", "--sensitive", "-f", "6", "qseqid", "sseqid", "length", "gapopen", "gaps", "btop"],
                  ["--ext", "banded-fast", "--more-sensitive"], ["--ext", "full", "--fast", "--comp-based-stats", "0", "-k", "3"], ["--ext", "full", "--id", "30", "--query-cover", "40"]):
        args = ["blastp", "-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.faa"), "-p", "4"] + extra
        _run([REF] + args + ["-o", str(tmp_path / "ref.tsv")])
        _run([CLI] + args + ["-o", str(tmp_path / "hip.tsv")])
        ref = open(tmp_path / "ref.tsv").read()
        assert len(ref.splitlines()) > 10, extra
        assert open(tmp_path / "hip.tsv").read() == ref, extra
        seen[" ".join(extra)] = ref
    assert len({seen["--ext banded-fast"], seen["--ext banded-slow"], seen["--ext full"]}) == 3


def _write_multi_hsp_files(d, seed=21):
    """Multi-domain queries (four domains) against targets with the domains in another order, in reverse order, one domain
    three times, two domains far apart, a single domain -- plus ordinary families: targets with one to four HSPs."""
    rng = np.random.default_rng(seed)
    db, doff, q, qoff = synth.generate(60, members=5, queries=60, seed=seed + 1, indel=0.03)
    seqs_t = [db[doff[i]:doff[i + 1]] for i in range(len(doff) - 1)]
    seqs_q = [q[qoff[i]:qoff[i + 1]] for i in range(len(qoff) - 1)]

    def mutate(s, rate):
        m = s.copy()
        mut = rng.random(len(m)) < rate
        m[mut] = rng.integers(0, 20, int(mut.sum()))
        return m

    def spacer():
        return rng.integers(0, 20, int(rng.integers(5, 90))).astype(np.int8)

    for k in range(40):
        doms = [rng.integers(0, 20, int(rng.integers(70, 160))).astype(np.int8) for _ in range(4)]
        seqs_q.append(np.concatenate([doms[0], spacer(), doms[1], spacer(), doms[2], spacer(), doms[3]]))
        seqs_t.append(np.concatenate([mutate(doms[2], 0.15), spacer(), mutate(doms[0], 0.2), spacer(), mutate(doms[1], 0.1)]))
        seqs_t.append(np.concatenate([mutate(doms[3], 0.1), spacer(), mutate(doms[2], 0.25), spacer(), mutate(doms[1], 0.2), spacer(), mutate(doms[0], 0.15)]))
        seqs_t.append(np.concatenate([mutate(doms[0], 0.1), spacer(), mutate(doms[0], 0.25), spacer(), mutate(doms[0], 0.3)]))
        seqs_t.append(np.concatenate([mutate(doms[1], 0.2), rng.integers(0, 20, 400).astype(np.int8), mutate(doms[3], 0.2)]))
        seqs_t.append(np.concatenate([spacer(), mutate(doms[1], 0.3)]))
    for name, seqs, prefix in (("db.faa", seqs_t, "t"), ("q.faa", seqs_q, "q")):
        off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])])
        synth.write_fasta(os.path.join(str(d), name), prefix, np.concatenate(seqs), off)
    qs = np.concatenate(seqs_q[60:80])
    qo = np.concatenate([[0], np.cumsum([len(s) for s in seqs_q[60:80]])])
    dna, off = synth.back_translate(qs, qo, seed=seed + 2)
    synth.write_dna_fasta(os.path.join(str(d), "reads.fna"), "r", dna, off)


def test_cli_max_hsps_matches_reference(tmp_path):
    """--max-hsps N (0 = all): every reported band of a target goes through round 2, the target's HSP list is culled by range
    overlap (Match::inner_culling), then alternative HSPs are searched on copies of the target with the found ranges masked
    (recompute_alt_hsps) -- per sensitivity, with the list cut at 2 / 3, with transcripts, in XML (Hit_num / Hsp_num), with the
    filters, over several reference blocks (an HSP list moves as one record group through the join), for translated queries."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    _write_multi_hsp_files(tmp_path)
    base = ["-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.faa"), "-p", "4"]
    per_target = {}
    strip = lambda t: "\n".join(l for l in t.splitlines() if "<BlastOutput_version>" not in l)
    for extra in (["--max-hsps", "0"], ["--max-hsps", "2"], ["--max-hsps", "3", "--fast"], ["--max-hsps", "0", "--sensitive"],
                  ["--max-hsps", "0", "-f", "6", "qseqid", "sseqid", "qstart", "qend", "sstart", "send", "evalue", "bitscore", "length", "gapopen", "btop"],
                  ["--max-hsps", "0", "-f", "5"], ["--max-hsps", "2", "-f", "0"],
                  ["--max-hsps", "0", "--id", "80"], ["--max-hsps", "0", "--query-cover", "20", "-k", "3"], ["--max-hsps", "0", "--comp-based-stats", "0"],
                  ["--max-hsps", "0", "-b0.00002"], ["--max-hsps", "0", "--top", "10"], ["--max-hsps", "0", "--ext", "full"],
                  ["--max-hsps", "0", "--masking", "0", "--algo", "1"]):
        _run([REF, "blastp"] + base + extra + ["-o", str(tmp_path / "ref.out")])
        _run([CLI, "blastp"] + base + extra + ["-o", str(tmp_path / "hip.out")])
        ref = open(tmp_path / "ref.out").read()
        assert len(ref) > 5000, extra
        assert strip(open(tmp_path / "hip.out").read()) == strip(ref), extra
        if "-f" not in extra:
            pairs = [tuple(l.split("\t")[:2]) for l in ref.splitlines()]
            per_target[" ".join(extra)] = max(pairs.count(p) for p in set(pairs))
    assert per_target["--max-hsps 0"] >= 4 and per_target["--max-hsps 2"] == 2 and per_target["--max-hsps 3 --fast"] == 3
    xargs = ["blastx", "-q", str(tmp_path / "reads.fna"), "-d", str(tmp_path / "db.faa"), "-p", "4", "--max-hsps", "0"]
    for extra in ([], ["--sensitive", "-f", "6", "qseqid", "sseqid", "qstart", "qend", "qframe", "sstart", "send", "evalue", "btop"]):
        _run([REF] + xargs + extra + ["-o", str(tmp_path / "ref.out")])
        _run([CLI] + xargs + extra + ["-o", str(tmp_path / "hip.out")])
        ref = open(tmp_path / "ref.out").read()
        pairs = [tuple(l.split("\t")[:2]) for l in ref.splitlines()]
        assert max(pairs.count(p) for p in set(pairs)) >= 3, extra
        assert open(tmp_path / "hip.out").read() == ref, extra


def test_cli_global_ranking_matches_reference(tmp_path):
    """--global-ranking N: the seed hits of every reference block only update a table of the N best targets per query (x-drop
    ungapped score over a target's seed hits); after the last block those targets are loaded as one block and extended over the
    full matrix. One and several reference blocks, three sensitivities, both algorithms, SEG, transcripts, --max-hsps, blastx."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    db, doff, q, qoff = synth.generate(150, members=12, queries=200, seed=31, decoy_frac=0.3)
    db, q = _plant_repeats(db, doff, np.random.default_rng(3)), _plant_repeats(q, qoff, np.random.default_rng(4))
    dna, off = synth.back_translate(q[: qoff[60]], qoff[:61], seed=32)
    synth.write_fasta(str(tmp_path / "db.faa"), "t", db, doff)
    synth.write_fasta(str(tmp_path / "q.faa"), "q", q, qoff)
    synth.write_dna_fasta(str(tmp_path / "reads.fna"), "r", dna, off)
    lines = {}
    for mode, query, extra in (("blastp", "q.faa", ["-g", "3"]), ("blastp", "q.faa", ["--global-ranking", "8", "-b0.00002"]), ("blastp", "q.faa", ["-g", "100"]),
                               ("blastp", "q.faa", ["-g", "4", "--fast", "-b0.00004", "-c1"]), ("blastp", "q.faa", ["-g", "5", "--sensitive", "-b0.00003"]),
                               ("blastp", "q.faa", ["-g", "3", "--algo", "1", "-b0.00003"]), ("blastp", "q.faa", ["-g", "3", "--masking", "seg", "-b0.00003"]),
                               ("blastp", "q.faa", ["-g", "3", "--masking", "0", "-b0.00003", "-f", "6", "qseqid", "sseqid", "qstart", "qend", "sstart", "send", "evalue", "bitscore", "btop"]),
                               ("blastp", "q.faa", ["-g", "2", "--max-hsps", "0", "-k", "1", "-b0.00003"]), ("blastp", "q.faa", ["-g", "6", "--ext", "full", "--top", "20", "-b0.00003"]),
                               ("blastx", "reads.fna", ["-g", "3", "-b0.00003"]), ("blastx", "reads.fna", ["-g", "2", "--sensitive"])):
        args = [mode, "-q", str(tmp_path / query), "-d", str(tmp_path / "db.faa"), "-p", "4"] + extra
        _run([REF] + args + ["-o", str(tmp_path / "ref.out")])
        _run([CLI] + args + ["-o", str(tmp_path / "hip.out")])
        ref = open(tmp_path / "ref.out").read()
        assert len(ref.splitlines()) > 50, extra
        assert open(tmp_path / "hip.out").read() == ref, (mode, extra)
        lines[" ".join([mode] + extra)] = len(ref.splitlines())
    assert lines["blastp -g 3"] < lines["blastp --global-ranking 8 -b0.00002"] < lines["blastp -g 100"]        # the table size limits what is extended
    r = subprocess.run([CLI, "blastp", "-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.faa"), "-o", str(tmp_path / "x.out"), "-g", "3", "--ext", "banded-fast"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "Global ranking only supports full matrix extension" in r.stderr


def test_cli_xml_format_matches_reference(tmp_path):
    """-f 5 (BLAST XML) for blastp (one and several reference blocks, queries without alignments) and blastx; the version line of the
    header names the program that wrote the file and is excluded."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    db, doff, q, qoff = synth.generate(200, members=10, queries=150, seed=51, decoy_frac=0.3)
    dna, off = synth.back_translate(q[: qoff[60]], qoff[:61], seed=52)
    synth.write_fasta(str(tmp_path / "db.faa"), "t", db, doff)
    synth.write_fasta(str(tmp_path / "q.faa"), "q", q, qoff)
    synth.write_dna_fasta(str(tmp_path / "reads.fna"), "r", dna, off)
    g = os.path.join(ROOT, "tests", "golden", "ref_ctest", "data.faa")
    strip = lambda t: "\n".join(l for l in t.splitlines() if "<BlastOutput_version>" not in l)
    for mode, query, database, extra in (("blastp", str(tmp_path / "q.faa"), str(tmp_path / "db.faa"), []), ("blastp", str(tmp_path / "q.faa"), str(tmp_path / "db.faa"), ["-b0.00003", "-k", "5"]),
                                         ("blastp", g, g, ["-k", "3", "--matrix", "blosum45"]), ("blastx", str(tmp_path / "reads.fna"), str(tmp_path / "db.faa"), ["-e", "1e-5"])):
        args = [mode, "-q", query, "-d", database, "-p", "4", "-f", "5"] + extra
        _run([REF] + args + ["-o", str(tmp_path / "ref.xml")])
        _run([CLI] + args + ["-o", str(tmp_path / "hip.xml")])
        ref = open(tmp_path / "ref.xml").read()
        assert ref.count("<Hsp>") > 100 and ref.count("<Iteration>") > 20 and ref.endswith("</BlastOutput>"), (mode, extra)
        assert strip(open(tmp_path / "hip.xml").read()) == strip(ref), (mode, extra)


def test_cli_daa_format_matches_reference(tmp_path):
    """-f 100 (DAA): byte-identical archives for blastp and blastx on one reference block (the header's build number is the reference
    version's); with several reference blocks the dictionary order is an artefact of the block loop, so there the reference's own
    `view` must print from our archive what it prints from its own; --salltitles / --sallseqid change the dictionary."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    db, doff, q, qoff = synth.generate(200, members=10, queries=150, seed=61, decoy_frac=0.3)
    dna, off = synth.back_translate(q[: qoff[60]], qoff[:61], seed=62)
    synth.write_fasta(str(tmp_path / "db.faa"), "t", db, doff)
    synth.write_fasta(str(tmp_path / "q.faa"), "q", q, qoff)
    synth.write_dna_fasta(str(tmp_path / "reads.fna"), "r", dna, off)
    g = os.path.join(ROOT, "tests", "golden", "ref_ctest", "data.faa")
    for mode, query, database, extra in (("blastp", str(tmp_path / "q.faa"), str(tmp_path / "db.faa"), []), ("blastx", str(tmp_path / "reads.fna"), str(tmp_path / "db.faa"), ["--sensitive"]),
                                         ("blastp", g, g, ["-k", "5", "--salltitles"]), ("blastp", g, g, ["-k", "2", "--sallseqid", "--matrix", "pam70"])):
        args = [mode, "-q", query, "-d", database, "-p", "1", "-f", "100"] + extra
        _run([REF] + args + ["-o", str(tmp_path / "ref")])
        _run([CLI] + args + ["-o", str(tmp_path / "hip")])
        ref, hip_ = open(tmp_path / "ref.daa", "rb").read(), open(tmp_path / "hip.daa", "rb").read()
        assert len(ref) > 20000, (mode, extra)
        assert hip_ == ref, (mode, extra, next(i for i in range(min(len(ref), len(hip_))) if ref[i] != hip_[i]) if ref[:len(hip_)] != hip_ else "length")
    args = ["blastp", "-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.faa"), "-p", "4", "-b0.00003", "-k", "7"]
    _run([CLI] + args + ["-f", "100", "-o", str(tmp_path / "blocks.daa")])
    _run([REF] + args + ["-o", str(tmp_path / "direct.tsv")])
    for fmt, name in ((["-f", "6"], "direct.tsv"),):
        _run([REF, "view", "--daa", str(tmp_path / "blocks.daa"), "-o", str(tmp_path / "view.tsv")] + fmt)
        want = open(tmp_path / name).read()
        assert len(want.splitlines()) > 300 and open(tmp_path / "view.tsv").read() == want


def test_cli_seg_masking_matches_reference(tmp_path):
    """--masking seg: the reference block hard-masked by SEG on the host (up front with the double-indexed algorithm, after the seed
    stage with the query-indexed one), queries untouched; with and without motif soft masking, several reference blocks, blastx."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    rng = np.random.default_rng(71)
    db, doff, q, qoff = synth.generate(300, members=10, queries=300, seed=71)
    db, q = _plant_repeats(db, doff, rng, frac=0.5), _plant_repeats(q, qoff, rng)
    dna, off = synth.back_translate(q[: qoff[80]], qoff[:81], seed=72)
    synth.write_fasta(str(tmp_path / "db.faa"), "t", db, doff)
    synth.write_fasta(str(tmp_path / "q.faa"), "q", q, qoff)
    synth.write_dna_fasta(str(tmp_path / "reads.fna"), "r", dna, off)
    seen = set()
    for mode, query, extra in (("blastp", "q.faa", []), ("blastp", "q.faa", ["--algo", "0"]), ("blastp", "q.faa", ["--algo", "1", "--motif-masking", "0"]),
                               ("blastp", "q.faa", ["--sensitive", "--algo", "0", "-b0.00004"]), ("blastp", "q.faa", ["--fast", "-b0.00004"]), ("blastx", "reads.fna", [])):
        args = [mode, "-q", str(tmp_path / query), "-d", str(tmp_path / "db.faa"), "-p", "4", "--masking", "seg"] + extra
        _run([REF] + args + ["-o", str(tmp_path / "ref.tsv")])
        r = _run([CLI] + args + ["-o", str(tmp_path / "hip.tsv")])
        assert "Masking reference (seg)" in r.stderr
        ref = open(tmp_path / "ref.tsv").read()
        assert len(ref.splitlines()) > 100, (mode, extra)
        assert open(tmp_path / "hip.tsv").read() == ref, (mode, extra)
        seen.add(ref)
    # SEG changes this workload: neither the unmasked nor the tantan-masked run gives the same lines
    base = ["blastp", "-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.faa"), "-p", "4", "--algo", "0"]
    for other in (["--masking", "0"], []):
        _run([REF] + base + other + ["-o", str(tmp_path / "other.tsv")])
        assert open(tmp_path / "other.tsv").read() not in seen


def test_cli_unlimited_target_seqs_matches_reference(tmp_path):
    """-k 0 = every target is reported (init_output: max_target_seqs = INT64_MAX), also over several reference blocks and with a filter."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    db, doff, q, qoff = synth.generate(60, members=60, queries=120, seed=81)          # families of 60: far more than 25 targets per query
    synth.write_fasta(str(tmp_path / "db.faa"), "t", db, doff)
    synth.write_fasta(str(tmp_path / "q.faa"), "q", q, qoff)
    for extra in (["-k", "0"], ["-k0", "-b0.0002", "--sensitive"], ["--max-target-seqs", "0", "--id", "35", "--fast"]):
        args = ["blastp", "-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.faa"), "-p", "4"] + extra
        _run([REF] + args + ["-o", str(tmp_path / "ref.tsv")])
        _run([CLI] + args + ["-o", str(tmp_path / "hip.tsv")])
        ref = open(tmp_path / "ref.tsv").read()
        per_query = {}
        for l in ref.splitlines():
            per_query[l.split("\t")[0]] = per_query.get(l.split("\t")[0], 0) + 1
        assert max(per_query.values()) > 25, extra
        assert open(tmp_path / "hip.tsv").read() == ref, extra


def test_cli_expert_options_match_reference(tmp_path):
    """--dbsize (effective database size of the e-values), --id2, --seed-cut, --gapped-filter-evalue, --stop-match-score: each against the
    reference, and each changes the default result of this workload."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    rng = np.random.default_rng(91)
    db, doff, q, qoff = synth.generate(300, members=10, queries=300, seed=91)
    db, q = _plant_repeats(db, doff, rng), _plant_repeats(q, qoff, rng)
    dna, off = synth.back_translate(q[: qoff[100]], qoff[:101], seed=92)
    synth.write_fasta(str(tmp_path / "db.faa"), "t", db, doff)
    synth.write_fasta(str(tmp_path / "q.faa"), "q", q, qoff)
    synth.write_dna_fasta(str(tmp_path / "reads.fna"), "r", dna, off)
    out = {}
    for name, mode, query, extra in (("plain", "blastp", "q.faa", []), ("dbsize", "blastp", "q.faa", ["--dbsize", "1000000000"]),
                                     ("id2", "blastp", "q.faa", ["--id2", "14"]), ("seed-cut", "blastp", "q.faa", ["--seed-cut", "0.5", "--masking", "0"]),
                                     ("plain sensitive", "blastp", "q.faa", ["--sensitive"]), ("gf off", "blastp", "q.faa", ["--sensitive", "--gapped-filter-evalue", "0"]),
                                     ("gf strict", "blastp", "q.faa", ["--sensitive", "--gapped-filter-evalue", "0.000001"]),
                                     ("plain x", "blastx", "reads.fna", []), ("stop score", "blastx", "reads.fna", ["--stop-match-score", "-4", "--min-orf", "1"])):
        args = [mode, "-q", str(tmp_path / query), "-d", str(tmp_path / "db.faa"), "-p", "1"] + extra
        daa = "-f" in extra
        _run([REF] + args + ["-o", str(tmp_path / ("ref" if daa else "ref.tsv"))])
        _run([CLI] + args + ["-o", str(tmp_path / ("hip" if daa else "hip.tsv"))])
        ref = open(tmp_path / ("ref.daa" if daa else "ref.tsv"), "rb").read()
        assert len(ref) > 5000, name
        assert open(tmp_path / ("hip.daa" if daa else "hip.tsv"), "rb").read() == ref, name
        out[name] = ref
    assert out["dbsize"] != out["plain"] and out["id2"] != out["plain"] and out["seed-cut"] != out["plain"]
    assert len({out["plain sensitive"], out["gf off"], out["gf strict"]}) >= 2


def test_cli_queries_from_standard_input(tmp_path):
    """No -q: the queries are read from standard input (gzip-compressed or not), blastp and blastx."""
    import gzip
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    db, doff, q, qoff = synth.generate(100, members=10, queries=80, seed=95)
    dna, off = synth.back_translate(q[: qoff[40]], qoff[:41], seed=96)
    synth.write_fasta(str(tmp_path / "db.faa"), "t", db, doff)
    synth.write_fasta(str(tmp_path / "q.faa"), "q", q, qoff)
    synth.write_dna_fasta(str(tmp_path / "reads.fna"), "r", dna, off)
    for mode, query in (("blastp", "q.faa"), ("blastx", "reads.fna")):
        _run([REF, mode, "-q", str(tmp_path / query), "-d", str(tmp_path / "db.faa"), "-p", "4", "-o", str(tmp_path / "ref.tsv")])
        ref = open(tmp_path / "ref.tsv").read()
        assert len(ref.splitlines()) > 50
        raw = open(tmp_path / query, "rb").read()
        for data in (raw, gzip.compress(raw)):
            r = subprocess.run([CLI, mode, "-d", str(tmp_path / "db.faa"), "-p", "4"], input=data, capture_output=True, timeout=300)
            assert r.returncode == 0, r.stderr[-1000:]
            assert r.stdout.decode() == ref, mode                    # no -o either: the alignments go to standard output


def _write_biased_files(d, seed=33):
    """Ordinary families plus families whose compositions are far from the matrix background -- two letters above 40 % of the
    sequence, hydrophobic-only and charged-only stretches, short queries -- so that NCBI's conditional test (the angle between the
    two compositions' deviations, the high-pair rule) takes both of its branches and the adjusted matrices differ from BLOSUM62."""
    rng = np.random.default_rng(seed)
    db, doff, q, qoff = synth.generate(150, members=6, queries=160, seed=seed + 1)
    seqs_t = [db[doff[i]:doff[i + 1]] for i in range(len(doff) - 1)]
    seqs_q = [q[qoff[i]:qoff[i + 1]] for i in range(len(qoff) - 1)]

    def mutate(s, rate, alphabet):
        m = s.copy()
        mut = rng.random(len(m)) < rate
        m[mut] = rng.choice(alphabet, int(mut.sum()))
        return m

    full = np.arange(20, dtype=np.int8)
    pools = [np.array([0, 10, 19, 9, 12, 13], np.int8),        # A L V I M F
             np.array([3, 6, 11, 1, 15, 5], np.int8),          # D E K R S Q
             np.array([7, 14, 15, 16, 2], np.int8)]            # G P S T N
    for k in range(90):
        n = int(rng.integers(40, 420))
        pool = pools[k % 3]
        weights = rng.dirichlet(np.ones(len(pool)) * (0.4 if k % 2 else 3.0))
        anc = np.where(rng.random(n) < 0.75, rng.choice(pool, n, p=weights), rng.choice(full, n)).astype(np.int8)
        seqs_q.append(mutate(anc, 0.2, full))
        for _ in range(3):
            seqs_t.append(mutate(anc, float(rng.uniform(0.1, 0.45)), pool if rng.random() < 0.5 else full))
        seqs_t.append(np.concatenate([rng.choice(full, int(rng.integers(10, 200))).astype(np.int8), mutate(anc, 0.25, full)]))
    for name, seqs, prefix in (("db.faa", seqs_t, "t"), ("q.faa", seqs_q, "q")):
        off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])])
        synth.write_fasta(os.path.join(str(d), name), prefix, np.concatenate(seqs), off)


def test_cli_comp_based_stats_matrix_adjust_matches_reference(tmp_path):
    """--comp-based-stats 2 .. 5 (row f4): per (query, target) a composition-adjusted scoring matrix (host double precision,
    cbs_adjust.cpp) and sweeps that score every DpTarget with its own matrix; byte-identical to the reference binary per mode,
    sensitivity, algorithm, with transcripts, several HSPs per target, several reference blocks, another standard matrix, --top and
    the filters. The reference's refusals are reproduced too."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    _write_biased_files(tmp_path)
    base = ["-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.faa"), "-p", "4"]
    outs = {}
    cases = [[m] + x for m in ("2", "3", "4", "5") for x in ([], ["--sensitive"], ["--fast"])]
    cases += [["4", "--masking", "0", "--algo", "0"], ["3", "--algo", "1"], ["5", "--max-hsps", "0"], ["4", "-b0.00004"],
              ["3", "-f", "6", "qseqid", "sseqid", "score", "qstart", "qend", "sstart", "send", "evalue", "bitscore", "length", "gapopen", "btop", "cigar"],
              ["4", "-f", "0"], ["5", "--matrix", "BLOSUM45"], ["4", "--matrix", "PAM70", "--top", "20"], ["2", "--id", "40", "-k", "5"],
              ["4", "--ext", "full"], ["5", "--more-sensitive", "-e", "10"],
              # Hauser bias AND adjusted matrices in full-matrix sweeps (the reference keeps the bias in every channel there)
              ["3", "--ext", "full"], ["2", "--max-hsps", "0"], ["3", "--max-hsps", "2", "-k", "3"]]
    refused = 0
    for extra in cases:
        args = ["--comp-based-stats"] + extra
        r = subprocess.run([REF, "blastp"] + base + args + ["-o", str(tmp_path / "ref.out")], capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            # the reference's full-matrix sweep cannot trace back on an adjusted matrix (dp/swipe/full_swipe.h:101-102): --ext full and
            # the alternative HSPs of --max-hsps end with this error as soon as such a target has an HSP to report -- so does this build
            assert "Traceback with adjusted matrix not supported" in r.stderr + r.stdout, (extra, r.stderr[-500:])
            r2 = subprocess.run([CLI, "blastp"] + base + args + ["-o", str(tmp_path / "hip.out")], capture_output=True, text=True, timeout=600)
            assert r2.returncode != 0 and "Traceback with adjusted matrix not supported" in r2.stderr + r2.stdout, (extra, r2.stderr[-500:])
            refused += 1
            continue
        _run([CLI, "blastp"] + base + args + ["-o", str(tmp_path / "hip.out")])
        ref = open(tmp_path / "ref.out").read()
        assert len(ref) > 5000, extra
        got = open(tmp_path / "hip.out").read()
        if got != ref:
            a, b = set(ref.splitlines()), set(got.splitlines())
            raise AssertionError("%s: %d lines only in the reference's output, %d only in ours; e.g. %s | %s" % (extra, len(a - b), len(b - a), sorted(a - b)[:2], sorted(b - a)[:2]))
        outs[" ".join(extra)] = ref
    assert refused >= 2                                      # --max-hsps 0 and --ext full
    # round 5: the matrices live for one pass over at most DMND_CBS_PASS_HITS seed hits; many small passes give the same text
    for extra in (["4"], ["3", "-f", "6", "qseqid", "sseqid", "score", "qstart", "qend", "sstart", "send", "evalue", "bitscore", "length", "gapopen", "btop", "cigar"]):
        r = subprocess.run([CLI, "blastp"] + base + ["--comp-based-stats"] + extra + ["-o", str(tmp_path / "passes.out")], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, DMND_CBS_PASS_HITS="300"))
        assert r.returncode == 0, r.stderr[-2000:]
        assert open(tmp_path / "passes.out").read() == outs[" ".join(extra)], extra
    # the modes are not the same computation: the adjusted matrices change scores
    _run([REF, "blastp"] + base + ["-o", str(tmp_path / "m1.out")])
    m1 = open(tmp_path / "m1.out").read()
    assert outs["4"] != m1 and outs["3"] != m1 and outs["5"] != outs["4"] and outs["3"] != outs["4"]
    # what the reference refuses (basic/config.cpp:688-700)
    for cmd in (["blastp", "--global-ranking", "5"], ):
        r1 = subprocess.run([REF] + cmd + base + ["--comp-based-stats", "3", "-o", str(tmp_path / "x.out")], capture_output=True, text=True)
        r2 = subprocess.run([CLI] + cmd + base + ["--comp-based-stats", "3", "-o", str(tmp_path / "y.out")], capture_output=True, text=True)
        assert r1.returncode != 0 and r2.returncode != 0
        assert "Global ranking is not supported in this mode." in r1.stderr + r1.stdout and "Global ranking is not supported in this mode." in r2.stderr + r2.stdout
    for val in ("6", "9"):
        r2 = subprocess.run([CLI, "blastp"] + base + ["--comp-based-stats", val, "-o", str(tmp_path / "y.out")], capture_output=True, text=True)
        assert r2.returncode != 0 and "Permitted values" in r2.stderr + r2.stdout


def test_cli_frameshift_matches_reference(tmp_path):
    """blastx -F 15 (row f4): reads with single-base insertions and deletions; the reference's legacy pipeline (ungapped ranking,
    score-only three-frame sweep + culling, traceback sweep, inner culling, global or range culling) over the three-frame sweep
    kernels. Byte-identical to the reference binary: the committed goldens and fresh runs in several sensitivities, -k, --top,
    --range-culling, --long-reads, filters, -F 5 / 30, one strand."""
    g = os.path.join(ROOT, "tests", "golden")
    base = ["blastx", "-q", os.path.join(g, "fs_reads.fna"), "-d", os.path.join(g, "fs_db.faa"), "-p", "4"]
    for extra, name in ((["-F", "15"], "fs_f15.tsv"), (["-F", "15", "-k", "3"], "fs_k3.tsv"), (["-F", "15", "--range-culling", "--top", "10"], "fs_f15_rc.tsv")):
        _run([CLI] + base + extra + ["-o", str(tmp_path / "hip.out")])
        want, got = open(os.path.join(g, name)).read(), open(tmp_path / "hip.out").read()
        if got != want:
            a, b = set(want.splitlines()), set(got.splitlines())
            raise AssertionError("%s: %d lines only in the golden, %d only in ours; e.g. %s | %s" % (name, len(a - b), len(b - a), sorted(a - b)[:2], sorted(b - a)[:2]))
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    db, doff, q, qoff = synth.generate(150, members=8, queries=220, seed=81)
    synth.write_fasta(str(tmp_path / "db.faa"), "t", db, doff)
    dna, off = synth.back_translate(q, qoff, seed=82)
    synth.write_dna_fasta(str(tmp_path / "reads.fna"), "r", *synth.indel_reads(dna, off, seed=83, deletion=0.004, insertion=0.004))
    base = ["blastx", "-q", str(tmp_path / "reads.fna"), "-d", str(tmp_path / "db.faa"), "-p", "4"]
    n_shift = 0
    for extra in (["-F", "15"], ["-F", "15", "-k", "2"], ["-F", "15", "-k", "1", "--sensitive"], ["-F", "15", "--top", "5"], ["--long-reads"],
                  ["-F", "15", "--range-culling", "-k", "2"], ["-F", "15", "--range-culling", "--range-cover", "20", "--top", "30"],
                  ["-F", "5", "--fast"], ["-F", "30", "--max-hsps", "0"], ["-F", "15", "--strand", "minus"], ["-F", "15", "--id", "60", "--query-cover", "30"],
                  ["-F", "15", "--masking", "0", "--algo", "1"], ["-F", "15", "-e", "1e-20", "--min-orf", "30"],
                  ["-F", "15", "-f", "6", "qseqid", "sseqid", "qstart", "qend", "qframe", "sstart", "send", "score", "length", "nident", "gaps", "qcovhsp", "qstrand", "qlen"]):
        _run([REF] + base + extra + ["-o", str(tmp_path / "ref.out")])
        _run([CLI] + base + extra + ["-o", str(tmp_path / "hip.out")])
        ref, got = open(tmp_path / "ref.out").read(), open(tmp_path / "hip.out").read()
        assert len(ref) > 1000, extra
        if got != ref:
            a, b = set(ref.splitlines()), set(got.splitlines())
            raise AssertionError("%s: %d lines only in the reference's output, %d only in ours; e.g. %s | %s" % (extra, len(a - b), len(b - a), sorted(a - b)[:2], sorted(b - a)[:2]))
    # the refusals (basic/config.cpp:822-825)
    for cmd, msg in ((["blastp", "-q", str(tmp_path / "db.faa"), "-d", str(tmp_path / "db.faa"), "-F", "15"], "Frameshift alignments are only supported for translated searches."),
                     (base + ["--range-culling"], "Query range culling is only supported in frameshift alignment mode (option -F).")):
        for binary in (REF, CLI):
            r = subprocess.run([binary] + cmd + ["-o", str(tmp_path / "x.out")], capture_output=True, text=True)
            assert r.returncode != 0 and msg in r.stderr + r.stdout, (binary, cmd)


def test_cli_frameshift_blocked_range_culling_matches_reference(tmp_path):
    """blastx -F with a database of several reference blocks (round 5, dmnd_join_blocks_range): the join of the per-block records
    uses the culler TargetCulling::get picks -- RangeCulling with --range-culling / --long-reads (a target is kept unless the part
    of the read it covers is covered already), GlobalCulling otherwise. Byte-identical to the reference binary with the same -b."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    db, doff, q, qoff = synth.generate(150, members=8, queries=220, seed=81)
    synth.write_fasta(str(tmp_path / "db.faa"), "t", db, doff)
    dna, off = synth.back_translate(q, qoff, seed=82)
    synth.write_dna_fasta(str(tmp_path / "reads.fna"), "r", *synth.indel_reads(dna, off, seed=83, deletion=0.004, insertion=0.004))
    base = ["blastx", "-q", str(tmp_path / "reads.fna"), "-d", str(tmp_path / "db.faa"), "-p", "4", "-b0.0001", "-c1"]
    differs = 0
    for extra in (["-F", "15", "--range-culling", "-k", "2"], ["--long-reads"], ["-F", "15", "--range-culling", "--range-cover", "20", "--top", "30"],
                  ["-F", "15", "--range-culling", "-k", "1", "--range-cover", "80"], ["-F", "15", "-k", "3"], ["-F", "15", "--top", "5"]):
        _run([REF] + base + extra + ["-o", str(tmp_path / "ref.out")])
        r = _run([CLI] + base + extra + ["-o", str(tmp_path / "hip.out")])
        assert "reference blocks=" in r.stderr
        ref, got = open(tmp_path / "ref.out").read(), open(tmp_path / "hip.out").read()
        assert len(ref) > 1000, extra
        if got != ref:
            a, b = set(ref.splitlines()), set(got.splitlines())
            raise AssertionError("%s: %d lines only in the reference's output, %d only in ours; e.g. %s | %s" % (extra, len(a - b), len(b - a), sorted(a - b)[:2], sorted(b - a)[:2]))
        if "--range-culling" in extra or "--long-reads" in extra:
            # the case is only a test of the range join if global culling of the same records would give another text
            _run([CLI] + base + [x for x in extra if x not in ("--range-culling", "--long-reads")] + (["-F", "15", "--top", "10"] if "--long-reads" in extra else []) + ["-o", str(tmp_path / "glob.out")])
            differs += open(tmp_path / "glob.out").read() != got
    assert differs >= 1


def test_cli_frameshift_alignment_fields_match_reference(tmp_path):
    """blastx -F 15 with the tabular fields that walk the alignment (round 4): btop, cigar, qseq_gapped, sseq_gapped, sseq and
    qseq_translated follow the alignment through its frame changes as the reference's HspContext::Iterator does; byte-identical
    to the reference binary on the reads with planted insertions and deletions."""
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure (run __graft_entry__.build() where /root/reference exists; oracle/_ref travels with gpurun)")
    g = os.path.join(ROOT, "tests", "golden")
    base = ["blastx", "-q", os.path.join(g, "fs_reads.fna"), "-d", os.path.join(g, "fs_db.faa"), "-p", "4", "-F", "15"]
    for extra in (["-f", "6", "qseqid", "sseqid", "qstart", "qend", "sstart", "send", "length", "btop", "cigar"],
                  ["-f", "6", "qseqid", "sseqid", "qseq_gapped", "sseq_gapped", "sseq", "qseq_translated", "qseq"],
                  ["--range-culling", "--top", "10", "-f", "6", "qseqid", "sseqid", "qframe", "btop", "qseq_translated"]):
        _run([REF] + base + extra + ["-o", str(tmp_path / "ref.out")])
        _run([CLI] + base + extra + ["-o", str(tmp_path / "hip.out")])
        ref, got = open(tmp_path / "ref.out").read(), open(tmp_path / "hip.out").read()
        assert len(ref) > 1000 and ("\\" in ref or "/" in ref or "qseq_gapped" not in extra), extra
        if got != ref:
            for i, (a, b) in enumerate(zip(ref.splitlines(), got.splitlines())):
                assert a == b, (extra, i, [(x[:80], y[:80]) for x, y in zip(a.split("\t"), b.split("\t")) if x != y][:2])
        assert got == ref, extra
    # -f 100: the archive of a -F run, byte for byte (one thread on both sides: the dictionary order follows the output order)
    args = ["blastx", "-q", os.path.join(g, "fs_reads.fna"), "-d", os.path.join(g, "fs_db.faa"), "-p", "1", "-F", "15", "-f", "100"]
    _run([REF] + args + ["-o", str(tmp_path / "ref")])
    _run([CLI] + args + ["-o", str(tmp_path / "hip")])
    ref, hip_ = open(tmp_path / "ref.daa", "rb").read(), open(tmp_path / "hip.daa", "rb").read()
    assert len(ref) > 20000
    assert hip_ == ref, next(i for i in range(min(len(ref), len(hip_))) if ref[i] != hip_[i]) if ref[:len(hip_)] != hip_[:len(ref)] else ("length", len(ref), len(hip_))
    # the pairwise format (coordinates move by single bases at a shift; every query without an alignment is listed: the legacy
    # pipeline is entered for each, align/align.cpp:167-171), XML and PAF
    strip = lambda t: "\n".join(l for l in t.splitlines() if "<BlastOutput_version>" not in l)
    for extra in (["-f", "0"], ["-f", "5"], ["-f", "paf"]):
        _run([REF] + base + extra + ["-o", str(tmp_path / "ref.out")])
        _run([CLI] + base + extra + ["-o", str(tmp_path / "hip.out")])
        ref, got = strip(open(tmp_path / "ref.out").read()), strip(open(tmp_path / "hip.out").read())
        assert len(ref) > 1000, extra
        if got != ref:
            for i, (a, b) in enumerate(zip(ref.splitlines(), got.splitlines())):
                assert a == b, (extra, i, a[:200], b[:200])
        assert got == ref, extra
