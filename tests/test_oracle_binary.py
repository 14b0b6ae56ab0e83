"""Pins the oracle BINARY (oracle/_ref/diamond = /root/reference compiled in place by oracle/Makefile) to the reference's own
committed ctest goldens: if it reproduces them, every golden minted from it (tests/golden/*.tap, *.tsv) and every A/B run
against it stands on the reference's word, not ours. CPU only; skipped where /root/reference does not exist (GPU box)."""
import os
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "diamond")
TD = "/root/reference/src/test"

pytestmark = pytest.mark.skipif(not (os.path.exists(REF) and os.path.isdir(TD)), reason="needs the reference tree and the built oracle binary")

# (golden file, arguments): CMakeLists.txt:455-572 of the reference
CASES = [
    ("blastp.out", ["blastp", "-q", TD + "/1.faa", "-d", TD + "/2.faa", "-p1"]),
    ("blastp-mid-sens.out", ["blastp", "-q", TD + "/3.faa", "-d", TD + "/4.faa", "--mid-sensitive", "-p1"]),
    ("blastp-f0.out", ["blastp", "-q", TD + "/1.faa", "-d", TD + "/2.faa", "-f0", "-p1"]),
    ("diamond-test-blastp-default.out", ["blastp", "-q", TD + "/data.faa", "-d", TD + "/data.faa", "-p1"]),
    ("diamond-test-blastp-blocked.out", ["blastp", "-q", TD + "/data.faa", "-d", TD + "/data.faa", "-c1", "-b0.00002", "-p4"]),
    ("diamond-test-blastp-query-indexed.out", ["blastp", "-q", TD + "/data.faa", "-d", TD + "/data.faa", "--more-sensitive", "-c1", "-p4", "--algo", "1"]),
    ("diamond-test-blastp-target-seqs.out", ["blastp", "-q", TD + "/data.faa", "-d", TD + "/data.faa", "-k3", "-c1", "-p4"]),
    ("diamond-test-blastp-comp-based-stats-0.out", ["blastp", "-q", TD + "/data.faa", "-d", TD + "/data.faa", "--more-sensitive", "-c1", "-p4", "--comp-based-stats", "0"]),
]


@pytest.mark.parametrize("golden,args", CASES, ids=[c[0] for c in CASES])
def test_oracle_binary_reproduces_reference_golden(tmp_path, golden, args):
    out = str(tmp_path / "out")
    r = subprocess.run([REF] + args + ["-o", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    want = open(os.path.join(TD, golden)).read()
    got = open(out).read()
    if "blocked" in golden or "-p4" in args:
        assert sorted(got.splitlines()) == sorted(want.splitlines())       # the ctest driver compares sorted output too
    else:
        assert got == want
