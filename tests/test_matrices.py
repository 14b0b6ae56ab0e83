"""dmnd_matrix_params: the standard scoring matrices of --matrix and the Gumbel constants of --gapopen / --gapextend
(host-only entry point, no device needed). Where the reference tree is present the generated table is re-derived from it."""
import importlib.util
import os

import numpy as np
import pytest

from diamond_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["blosum45", "blosum50", "blosum62", "blosum80", "blosum90", "pam30", "pam70", "pam250"]
DEFAULT_GAPS = {"blosum45": (14, 2), "blosum50": (13, 2), "blosum62": (11, 1), "blosum80": (10, 1), "blosum90": (10, 1),
                "pam30": (9, 1), "pam70": (10, 1), "pam250": (14, 2)}          # NCBI's defaults, src/stats/matrices/*.h


def _tool():
    spec = importlib.util.spec_from_file_location("make_matrix_tables", os.path.join(ROOT, "tools", "make_matrix_tables.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("name", NAMES)
def test_matrix_shape_and_defaults(name):
    p = hip.matrix_params(name.upper())                 # names are case-insensitive (StandardMatrix::get, stats.cpp:59-66)
    m = hip.matrix_of(p).reshape(32, 32).astype(int)
    assert (p.gap_open, p.gap_extend) == DEFAULT_GAPS[name]
    assert np.array_equal(m[:25, :25], m[:25, :25].T)
    assert (m[26:, :] == -128).all() and (m[:, 26:] == -128).all()
    assert (np.diag(m)[:20] > 0).all() and m[24, 24] == 1                      # stop codon against itself: stop_match_score
    assert (m[25, :26] == m[:26, :26].min()).all()                             # hard-mask letter: the lowest score everywhere
    assert 0.1 < p.lambda_ < 0.4 and 0 < p.K < 0.2 and p.alpha > p.u_alpha > 0 and p.alpha_v > p.u_alpha_v > 0 and p.sigma >= p.alpha_v


def test_blosum62_rows_are_the_published_constants():
    p = hip.matrix_params("blosum62", 11, 1)
    assert (p.lambda_, p.K, p.alpha, p.alpha_v, p.sigma) == (0.267, 0.041, 1.9, 42.6028, 43.6362)
    assert (p.u_alpha, p.u_alpha_v) == (0.7916, 4.96466)
    q = hip.matrix_params("blosum62", 9, 2)
    assert (q.gap_open, q.gap_extend, q.lambda_, q.K) == (9, 2, 0.279, 0.058)
    assert np.array_equal(hip.matrix_of(q), hip.matrix_of(p))
    d = hip.default_params()
    assert (d.lambda_, d.K, d.gap_open, d.gap_extend, d.max_evalue) == (0.267, 0.041, 11, 1, 0.001)


def test_errors_are_the_reference_messages():
    with pytest.raises(hip.DiamondHipError, match="Unknown scoring matrix: blosum61"):
        hip.matrix_params("blosum61")
    with pytest.raises(hip.DiamondHipError, match="Gap penalty settings are outside the supported range"):
        hip.matrix_params("blosum62", 5, 5)
    with pytest.raises(hip.DiamondHipError, match="outside the supported range"):
        hip.matrix_params("pam30", 11, 1)
    keep = hip.default_params()
    keep.db_letters = 123.0
    keep.max_evalue = 10.0
    p = hip.matrix_params("pam70", params=keep)
    assert (p.db_letters, p.max_evalue) == (123.0, 10.0)                       # left as they were


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/stats/matrices"), reason="reference tree not present")
@pytest.mark.parametrize("name", NAMES)
def test_table_against_the_reference_tree(name):
    d_open, d_ext, rows, scores = _tool().parse("/root/reference/src/stats/matrices/%s.h" % name)
    scores = np.array(scores).reshape(26, 26)
    assert len(rows) >= 5
    for (go, ge, lam, k, alpha, alpha_v, sigma) in rows[1:]:
        p = hip.matrix_params(name, go, ge)
        assert np.array_equal(hip.matrix_of(p).reshape(32, 32)[:26, :26], scores)
        assert (p.lambda_, p.K, p.alpha, p.alpha_v, p.sigma) == (float(lam), float(k), float(alpha), float(alpha_v), float(sigma))
        assert (p.u_alpha, p.u_alpha_v) == (float(rows[0][4]), float(rows[0][5]))
    assert (d_open, d_ext) == DEFAULT_GAPS[name]


def test_masking_lambda_follows_the_reference_calculator():
    """Masking::Masking (masking.cpp:140-143): 0.324032 for BLOSUM62 (the reference's own comment), the valid root for the others,
    -1 for PAM250, whose only roots imply letter probabilities outside [0, 1]."""
    import ctypes
    lib = hip.load()
    lam = {n: lib.dmnd_masking_lambda(ctypes.byref(hip.matrix_params(n))) for n in NAMES}
    assert round(lam["blosum62"], 6) == 0.324032
    assert lam["pam250"] == -1.0
    want = {"blosum45": 0.233472, "blosum50": 0.236973, "blosum80": 0.353941, "blosum90": 0.341771, "pam30": 0.346637, "pam70": 0.342098}
    for n, v in want.items():
        assert abs(lam[n] - v) < 1e-6, n
        m = hip.matrix_of(hip.matrix_params(n)).reshape(32, 32)[:20, :20].astype(float)
        inv = np.linalg.inv(np.exp(lam[n] * m))
        assert abs(inv.sum() - 1.0) < 1e-9 and (inv.sum(0) >= 0).all() and (inv.sum(1) <= 1).all()
