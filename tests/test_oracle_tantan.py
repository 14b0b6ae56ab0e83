"""Pins the tantan restatement (oracle/tantan.c) and the CPU lane emulator of the HIP masking kernel
(tests/emu/mask_emu.cpp over diamond_amd/csrc/mask_core.h) against the genuine reference, tapped at Util::tantan::mask
(tests/golden/tantan.tap: every query and target of the reference's own fixture, before and after masking, plus the
float likelihood-ratio matrix). Bit-exact float arithmetic: the masked positions must be identical. CPU only."""
import os
import numpy as np
import pytest

import oracle_py as orc
import emu_py as emu
from tapfile import read_tantan_tap
from test_oracle_seed import blosum62_matrix8

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def tap():
    return read_tantan_tap(os.path.join(GOLDEN, "tantan.tap"))


def test_likelihood_ratio_matrix_equals_reference(tap):
    hdr, _ = tap
    lr = orc.tantan_matrix(blosum62_matrix8())
    assert np.array_equal(lr[:26, :26], hdr["lr"][:26, :26])                 # float32 bit patterns
    assert orc.tantan_lambda(blosum62_matrix8()) == pytest.approx(0.324032, abs=1e-6)   # the value the reference's source quotes
    assert (hdr["p_repeat"], hdr["p_repeat_end"], hdr["p_mask"]) == (np.float32(0.005), np.float32(0.05), np.float32(0.9))


def test_masked_positions_equal_reference(tap):
    hdr, recs = tap
    p = emu.tantan_params()
    masked = touched = 0
    for r in recs:
        assert r["mask_mode"] == 1
        want = r["after"]
        got, n = orc.tantan_mask(r["before"], hdr["lr"], hdr["p_repeat"], hdr["p_repeat_end"], hdr["repeat_growth"], hdr["p_mask"])
        assert np.array_equal(got, want)
        got2, n2 = emu.tantan_mask(p, hdr["lr"], r["before"])
        assert np.array_equal(got2, want) and n2 == n
        k = int((want != r["before"]).sum())
        masked += k
        touched += k > 0
    assert len(recs) > 700 and masked > 3000 and touched >= 20


def test_edge_lengths_and_repeats():
    """Lengths around the 16-step rescaling and 64-lane chunk boundaries, periodic and homopolymer sequences."""
    lr = orc.tantan_matrix(blosum62_matrix8())
    p = emu.tantan_params()
    rng = np.random.default_rng(9)
    for n in [1, 2, 15, 16, 17, 49, 50, 51, 63, 64, 65, 127, 128, 129, 300, 1000]:
        for kind in range(3):
            if kind == 0:
                s = rng.integers(0, 20, n).astype(np.int8)
            elif kind == 1:
                unit = rng.integers(0, 20, int(rng.integers(1, 8))).astype(np.int8)
                s = np.resize(unit, n).astype(np.int8)
                flip = rng.random(n) < 0.1
                s[flip] = rng.integers(0, 20, int(flip.sum()))
            else:
                s = rng.integers(0, 20, n).astype(np.int8)
                a = int(rng.integers(0, max(1, n - 20)))
                s[a:a + 30] = s[a]
            a1, n1 = orc.tantan_mask(s, lr)
            a2, n2 = emu.tantan_mask(p, lr, s)
            assert np.array_equal(a1, a2) and n1 == n2, (n, kind)
