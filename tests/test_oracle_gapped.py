"""Pins the gapped-filter restatement (oracle/gapped_filter.c) against the genuine reference, tapped at
Extension::gapped_filter (tests/golden/gf_sensitive.tap): surviving target sets and both CutoffTable2D values. Then checks
the CPU lane emulator of the HIP kernel (tests/emu/gapped_emu.cpp over diamond_amd/csrc/gapped_core.h) against the oracle
hit by hit. CPU only."""
import os
import numpy as np
import pytest

import oracle_py as orc
import emu_py as emu
from tapfile import read_gf_tap, read_ext_tap
from test_oracle_seed import blosum62_matrix8

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def tap():
    recs = read_gf_tap(os.path.join(GOLDEN, "gf_sensitive.tap"))
    cfg, _ = read_ext_tap(os.path.join(GOLDEN, "ext_sensitive.tap"), max_records=1)
    return recs, cfg, blosum62_matrix8()


def _seqs(cfg, r, t):
    qd, td, tl = cfg["query"]["data"], cfg["target"]["data"], cfg["target"]["limits"]
    return qd[r["query_offset"]:r["query_offset"] + r["qlen"]], td[tl[t["block_id"]]:tl[t["block_id"] + 1] - 1]


def test_surviving_targets_and_cutoffs_equal_reference(tap):
    recs, cfg, m8 = tap
    r0 = recs[0]
    assert (r0["gapped_filter_evalue"], r0["gapped_filter_evalue1"], r0["window"], r0["diag_score"]) == (1.0, 2000.0, 200, 20)
    t1, t2 = orc.cutoff_table2d(r0["gapped_filter_evalue1"]), orc.cutoff_table2d(r0["gapped_filter_evalue"])
    n = dropped = 0
    for r in recs:
        out = set(r["out"].tolist())
        for t in r["targets"]:
            q, s = _seqs(cfg, r, t)
            b1, b2 = int(len(q)).bit_length(), int(len(s)).bit_length()
            assert (t1[b1, b2], t2[b1, b2]) == (t["cutoff1"], t["cutoff2"])
            got = orc.gapped_filter_target(m8, q, r["cbs"], s, t["hits"], t["cutoff1"], t["cutoff2"], r["window"], r["diag_score"])
            assert got == (t["block_id"] in out), (r["query_offset"], t["block_id"])
            n += 1
            dropped += not got
    assert n > 2000 and dropped > 20


def test_emulated_kernel_equals_oracle_per_hit(tap):
    recs, cfg, m8 = tap
    p = emu.GfParams(diag_score=20, gap_open=11, gap_extend=1, window2=200, use_cbs=1, contexts=1)
    n = stage2 = 0
    for r in recs[::3]:
        for t in r["targets"]:
            q, s = _seqs(cfg, r, t)
            for hi, hj in t["hits"][:, :2]:
                f1 = orc.gapped_filter_hit(m8, q, r["cbs"], s, hi, hj, 64, 100, 20)
                flag, e1, e2 = emu.gapped_filter_hit(p, m8, q, r["cbs"], s, hi, hj, t["cutoff1"], t["cutoff2"])
                assert e1 == f1
                if f1 > t["cutoff1"]:
                    f2 = orc.gapped_filter_hit(m8, q, r["cbs"], s, hi, hj, 128, 200, 20)
                    assert e2 == f2 and flag == (f2 > t["cutoff2"])
                    stage2 += 1
                else:
                    assert e2 == -1 and flag == 0
                n += 1
    assert n > 1000 and stage2 > 500


def test_scan_edge_geometry_against_oracle():
    """Hits near sequence ends, short sequences, bias saturation: band/window clipping and profile padding."""
    rng = np.random.default_rng(5)
    m8 = blosum62_matrix8()
    p = emu.GfParams(diag_score=20, gap_open=11, gap_extend=1, window2=200, use_cbs=1, contexts=1)
    for _ in range(300):
        qlen, slen = int(rng.integers(1, 400)), int(rng.integers(1, 400))
        q = rng.integers(0, 20, qlen).astype(np.int8)
        s = rng.integers(0, 25, slen).astype(np.int8)
        k = min(qlen, slen)
        if rng.random() < 0.7:                                   # plant a conserved stretch so that scores matter
            a, b = int(rng.integers(0, qlen - k + 1)), int(rng.integers(0, slen - k + 1))
            s[b:b + k] = q[a:a + k]
        cbs = rng.integers(-128, 128, qlen).astype(np.int8) if rng.random() < 0.2 else rng.integers(-3, 4, qlen).astype(np.int8)
        hi, hj = int(rng.integers(0, qlen)), int(rng.integers(0, slen))
        c1, c2 = int(rng.integers(0, 60)), int(rng.integers(0, 80))
        f1 = orc.gapped_filter_hit(m8, q, cbs, s, hi, hj, 64, 100, 20)
        f2 = orc.gapped_filter_hit(m8, q, cbs, s, hi, hj, 128, 200, 20)
        flag, e1, e2 = emu.gapped_filter_hit(p, m8, q, cbs, s, hi, hj, c1, c2)
        assert e1 == f1
        assert (e2, flag) == ((f2, int(f2 > c2)) if f1 > c1 else (-1, 0))
