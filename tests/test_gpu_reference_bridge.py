"""-m gpu A/B test at the reference's own operator seams: oracle/_ref/diamond_hip is the GENUINE reference
(compiled in place from /root/reference by oracle/Makefile) whose DP::BandedSwipe::swipe is answered by our
C ABI on the MI355X (oracle/ref_hip_bridge.cpp) -- and, with DMND_BRIDGE_SEED=1, whose Search::search_shape
(the seed stage's dispatch point) is answered by dmnd_seed_search as well, the hits going into the
reference's own HitBuffer. Its output must be byte-identical to the unmodified reference binary on the same
inputs -- hit sets, scores, coordinates, identities, CIGARs, e-values."""
import os
import subprocess
import pytest

from diamond_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "diamond")
BRIDGE = os.path.join(ROOT, "oracle", "_ref", "diamond_hip")


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    if not (os.path.exists(REF) and os.path.exists(BRIDGE)):
        pytest.fail("oracle/_ref binaries are missing: under -m gpu the bridged reference is the checker, its absence is a failure (build where /root/reference exists)")
    d = tmp_path_factory.mktemp("bridge")
    db, doff, q, qoff = synth.generate(300, members=10, queries=300, seed=11)
    synth.write_fasta(str(d / "db.faa"), "t", db, doff)
    synth.write_fasta(str(d / "q.faa"), "q", q, qoff)
    return d


def _run(binary, d, name, extra, seed_seam=False, algo=("--algo", "0")):
    env = dict(os.environ, DMND_HIP_LIB=os.path.join(ROOT, "diamond_amd", "libdiamond_hip.so"))
    if seed_seam:
        env["DMND_BRIDGE_SEED"] = "1"
    out = d / name
    cmd = [binary, "blastp", "-q", str(d / "q.faa"), "-d", str(d / "db.faa"), "-o", str(out), "-p", "4"] + list(algo) + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "outside the bridge" not in r.stderr, r.stderr[-2000:]      # the seed seam really answered (no silent fall-back to the reference's own seed stage)
    return open(out).read()


@pytest.mark.parametrize("name,extra", [
    ("fast", ["--fast"]),
    ("fast_cigar", ["--fast", "-f", "6", "qseqid", "sseqid", "score", "qstart", "qend", "sstart", "send", "cigar", "evalue", "bitscore"]),
    ("default", []),
    ("sensitive_k5", ["--sensitive", "-k", "5"]),
    # --comp-based-stats 3 / 4 / 5: the reference makes the per-target adjusted matrices, DpTarget::matrix crosses the seam
    # (dmnd_host_target::matrix) and the device sweeps every target with its own
    ("cbs3", ["--comp-based-stats", "3"]),
    ("cbs4_cigar", ["--comp-based-stats", "4", "-f", "6", "qseqid", "sseqid", "score", "qstart", "qend", "sstart", "send", "cigar", "evalue", "bitscore"]),
    ("cbs5_sensitive", ["--comp-based-stats", "5", "--sensitive"]),
])
def test_reference_with_hip_swipe_is_byte_identical(data, name, extra):
    ref = _run(REF, data, name + ".ref.tsv", extra)
    got = _run(BRIDGE, data, name + ".hip.tsv", extra)
    assert len(ref.splitlines()) > 200
    assert got == ref


@pytest.mark.parametrize("name,extra,algo", [
    ("fast", ["--fast"], ("--algo", "0")),
    ("fast_nomask", ["--fast", "--masking", "0", "--motif-masking", "0"], ("--algo", "0")),
    ("default", [], ("--algo", "0")),
    ("sensitive", ["--sensitive"], ("--algo", "0")),
    ("default_c1", ["-c", "1"], ("--algo", "0")),
    ("fast_query_indexed", ["--fast"], ("--algo", "1")),
])
def test_reference_with_both_seams_is_byte_identical(data, name, extra, algo):
    """Search::search_shape AND DP::BandedSwipe::swipe inside the genuine reference answered by libdiamond_hip.so: default
    masking (tantan on both blocks before the seed stage, motif soft masking during seed enumeration) unless switched off."""
    ref = _run(REF, data, name + ".ref2.tsv", extra, algo=algo)
    got = _run(BRIDGE, data, name + ".hip2.tsv", extra, seed_seam=True, algo=algo)
    assert len(ref.splitlines()) > 200
    assert got == ref
