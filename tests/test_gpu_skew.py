"""-m gpu: a workload that takes the seed stage's slow paths by itself (round 5). Every other full-size run is on families of 10
members with i.i.d. background letters; here the database is 40 families of 6000 members (a query's seeds join thousands of
reference positions: the joined-position lists outgrow their first buffer and phase 1 runs again, the lists are sorted by seed and
filtered by the LDS-tiled kernel -- the situation the reference's 1024 x 1024 stage-1 tiles exist for, search/hamming/kernel.h:29-50,
basic/config.cpp:423), with tandem repeats planted into a third of the sequences. Byte-identical A/B against the reference binary for
--fast without masking and the default sensitivity with tantan; that the slow paths were really taken is read from the library's DMND_TRACE lines."""
import os
import re
import subprocess
import numpy as np
import pytest

from diamond_amd import synth

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1500)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "diamond")
CLI = os.path.join(ROOT, "diamond_amd", "diamond-hip")
THREADS = str(min(16, os.cpu_count() or 8))


def _plant_repeats(data, off, rng, frac=0.3):
    data = data.copy()
    for i in np.flatnonzero(rng.random(len(off) - 1) < frac):
        n = int(off[i + 1] - off[i])
        if n < 80:
            continue
        unit = rng.integers(0, 20, int(rng.integers(1, 7))).astype(data.dtype)
        a, L = int(rng.integers(0, n - 60)), int(rng.integers(25, 60))
        seg = np.resize(unit, L)
        flip = rng.random(L) < 0.05
        seg[flip] = rng.integers(0, 20, int(flip.sum()))
        data[off[i] + a: off[i] + a + L] = seg
    return data


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/diamond is missing: under -m gpu the reference binary is the checker, its absence is a failure")
    d = tmp_path_factory.mktemp("skew")
    rng = np.random.default_rng(7)
    # (measured while calibrating: at the generator's default divergence 4000 queries join 1.7e6 positions of this database, not enough;
    #  members 10-30 % from their ancestor and queries 10-30 % from a member join ~14 reference positions per query position)
    db, doff, q, qoff = synth.generate(40, members=6000, queries=3000, seed=20260924, sub=(0.1, 0.3), qsub=(0.1, 0.3))
    assert len(doff) - 1 >= 200_000
    synth.write_fasta(str(d / "db.faa"), "t", _plant_repeats(db, doff, rng), doff)
    synth.write_fasta(str(d / "q.faa"), "q", _plant_repeats(q, qoff, rng), qoff)
    r = subprocess.run([REF, "makedb", "--in", str(d / "db.faa"), "-d", str(d / "db"), "-p", THREADS], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1000:]
    return d


@pytest.mark.parametrize("flags", [["--fast", "--masking", "0"], []], ids=["fast_unmasked", "default_tantan"])
def test_skewed_families_are_byte_identical_and_take_the_slow_paths(files, flags):
    d = files
    tag = "_".join(x.strip("-") for x in flags) or "default"
    common = ["blastp", "--algo", "0", "--motif-masking", "0", "-q", str(d / "q.faa"), "-d", str(d / "db.dmnd"), "-p", THREADS] + flags
    r = subprocess.run([REF] + common + ["-o", str(d / (tag + "_ref.tsv"))], capture_output=True, text=True, timeout=1400)
    assert r.returncode == 0, r.stderr[-2000:]
    h = subprocess.run([CLI] + common + ["-o", str(d / (tag + "_hip.tsv"))], capture_output=True, text=True, timeout=1400, env=dict(os.environ, DMND_TRACE="1"))
    assert h.returncode == 0, h.stderr[-2000:]
    a, b = open(d / (tag + "_ref.tsv"), "rb").read(), open(d / (tag + "_hip.tsv"), "rb").read()
    assert a.count(b"\n") > 60_000
    if a != b:
        sa, sb = set(a.decode().splitlines()), set(b.decode().splitlines())
        raise AssertionError("%s: %d lines only in the reference, %d only in diamond-hip, e.g. %s | %s" % (tag, len(sa - sb), len(sb - sa), sorted(sa - sb)[:3], sorted(sb - sa)[:3]))
    # the slow paths, taken because of the data and not because an environment variable forced them
    joined = [int(x) for m in re.finditer(r"joined reference positions per shape:((?: \d+)+)", h.stderr) for x in m.group(1).split()]
    assert joined and max(joined) >= 1 << 22, joined[:8]
    assert "tiled pair filter" in h.stderr, joined[:8]
    assert "phase 1 runs again" in h.stderr, joined[:8]
