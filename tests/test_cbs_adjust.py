"""Composition-based matrix adjustment (--comp-based-stats 2..5) on the host, against known answers tapped from the genuine
reference at Stats::adjust_matrix and Stats::TargetMatrix::TargetMatrix (/root/reference/src/stats/cbs.cpp:94-173; fixtures:
tests/golden/cbs_*.tap, minted by tests/golden/make_cbs_golden.sh). The adjusted matrices are rounded integers out of a
double-precision Newton iteration: they have to equal the reference's entry for entry."""
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import tapfile  # noqa: E402
from diamond_amd import hip  # noqa: E402

CASES = [("cbs_mode3.tap", "blosum62", 3), ("cbs_mode4.tap", "blosum62", 4), ("cbs_mode5.tap", "blosum62", 5),
         ("cbs_blosum45.tap", "blosum45", 5), ("cbs_pam70.tap", "pam70", 4)]


@pytest.mark.parametrize("name,matrix,mode", CASES)
def test_rules_and_matrices_equal_the_reference(name, matrix, mode):
    hdr, adj, tmx = tapfile.read_cbs_tap(os.path.join(HERE, "golden", name))
    p = hip.matrix_params(matrix)
    assert hdr["cbs"] == mode and hdr["scale"] == 1
    assert np.array_equal(hip.matrix_of(p), hdr["matrix8"])
    # ScoreMatrix::ideal_lambda: the same double
    assert hip.load().dmnd_cbs_ideal_lambda(p) == hdr["ideal_lambda"]
    assert len(adj) > 100 and len(tmx) > 100
    rules = set()
    for a in adj:
        comp, n = hip.cbs_composition(a["target"])
        assert hip.cbs_rule(p, mode, a["query_comp"], a["query_len"], a["target"]) == a["rule"]
        rules.add(a["rule"])
    assert rules == ({-1, 4} if mode in (2, 3) else {4} if mode == 4 else {0, 4})
    for x in tmx:
        got = hip.cbs_target_matrix(p, x["rule"], x["query_comp"], x["query_len"], x["target"])
        assert np.array_equal(got[:26, :26], x["scores"][:, :26]), (x["rule"], x["query_len"], len(x["target"]))
        # letters above 25 never score (the standard table's convention)
        assert (got[26:, :] == -128).all() and (got[:, 26:] == -128).all()
        # every entry outside the residue + mask-letter block keeps the standard score
        std = hip.matrix_of(p)
        keep = np.ones((26, 26), bool)
        idx = list(range(20)) + [23]
        keep[np.ix_(idx, idx)] = False
        assert np.array_equal(got[:26, :26][keep], std[:26, :26][keep])
        assert x["score_min"] == max(int(x["scores"][np.ix_(idx, idx)].min()), -128)


def test_composition_counts_residues_only():
    seq = np.array([0, 1, 1, 23, 24, 19, 25, 31, 0x80], np.uint8).view(np.int8)      # A R R X * V (hard mask) (delimiter) masked A
    comp, n = hip.cbs_composition(seq)
    assert n == 5
    want = np.zeros(20)
    want[0], want[1], want[19] = 2 / 5, 2 / 5, 1 / 5
    assert np.array_equal(comp, want)
    comp, n = hip.cbs_composition(np.array([23, 23], np.int8))
    assert n == 0 and not comp.any()


def test_degenerate_pairs():
    p = hip.default_params()
    comp, n = hip.cbs_composition(np.arange(20, dtype=np.int8))
    # empty target or a query without residues: no adjustment (cbs.cpp:95)
    assert hip.cbs_rule(p, 4, comp, 0, np.arange(20, dtype=np.int8)) == -1
    assert hip.cbs_rule(p, 4, comp, n, np.zeros(0, np.int8)) == -1
    assert hip.cbs_rule(p, 1, comp, n, np.arange(20, dtype=np.int8)) == -1
    # a target of mask letters only: its composition is all zero, the pseudocounts make it the background
    m = hip.cbs_target_matrix(p, 4, comp, n, np.full(30, 23, np.int8))
    assert m.shape == (32, 32) and m[0, 0] > 0 and m[23, 23] == -1
    # a custom matrix has no joint probabilities
    q = hip.default_params()
    q.matrix8[0] = 9
    with pytest.raises(hip.DiamondHipError):
        hip.cbs_rule(q, 4, comp, n, np.arange(20, dtype=np.int8))


def test_tables_are_the_reference_literals():
    """cbs_tables.h carries the reference's decimal literals as text (tools/make_cbs_tables.py): regenerate and compare where the
    reference tree exists."""
    if not os.path.isdir("/root/reference/src/stats/matrices"):
        pytest.skip("/root/reference not present")
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "cbs_tables.h")
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_cbs_tables.py"), "/root/reference", out], stdout=subprocess.DEVNULL)
        assert open(out).read() == open(os.path.join(ROOT, "diamond_amd", "csrc", "cbs_tables.h")).read()
