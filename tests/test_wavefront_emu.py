"""CPU check of the HIP kernel's per-lane code (diamond_amd/csrc/swipe_core.h), run through the
64-lane emulator tests/emu/swipe_emu.cpp, against the oracle: the anti-diagonal wavefront schedule
must reproduce the reference's column sweep bit for bit (scores, end/start cells, transcripts)."""
import os
import numpy as np
import pytest

import oracle_py as orc
import emu_py as emu
from tapfile import read_tap

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KEYS = "score q_begin q_end s_begin s_end length identities mismatches positives gap_openings gaps transcript_len".split()


def _compare(q, cbs, t, d0, d1, M, go, ge, force_p=0):
    for mode, omode in ((0, orc.SCORE_ONLY), (1, orc.COORDS), (2, orc.TRACEBACK)):
        rc, o, otr = orc.banded_swipe(q, cbs, t, d0, d1, M, go, ge, omode)
        assert rc == 0
        rc, e, etr = emu.banded_swipe(q, cbs, t, d0, d1, M, go, ge, mode, force_p)
        assert rc == 0 and e["status"] == 0
        assert e["score"] == o["score"]
        if o["score"] <= 0:
            continue
        if mode >= 1:
            assert (e["q_end"], e["s_end"]) == (o["q_end"], o["s_end"])
        if mode == 2:
            for k in KEYS:
                assert e[k] == o[k], (k, e, o)
            assert np.array_equal(etr, otr)


@pytest.mark.parametrize("tap", ["swipe_default.tap", "swipe_fast.tap", "swipe_blastx.tap"])
def test_emulated_wavefront_on_reference_targets(tap):
    hdr, recs = read_tap(os.path.join(GOLDEN, tap))
    n = 0
    for rec in recs[::3]:
        for t in rec["targets"]:
            _compare(rec["query"], rec["cbs"], t["seq"], t["d_begin"], t["d_end"], hdr["matrix8"], hdr["gap_open"], hdr["gap_extend"])
            n += 1
    assert n >= 5


def test_emulated_wavefront_random_geometry():
    hdr, _ = read_tap(os.path.join(GOLDEN, "swipe_fast.tap"), max_records=1)
    M = hdr["matrix8"]
    rng = np.random.default_rng(5)
    for it in range(300):
        qlen = int(rng.integers(1, 200))
        tlen = int(rng.integers(1, 200))
        q = rng.integers(0, 25, qlen).astype(np.int8)
        if it % 2 == 0:     # related pair with indels so gapped paths and ties occur
            t = q.copy()
            mut = rng.random(qlen) < 0.3
            t[mut] = rng.integers(0, 20, int(mut.sum()))
            cut = int(rng.integers(0, qlen))
            t = np.concatenate([t[:cut], rng.integers(0, 20, int(rng.integers(0, 6))).astype(np.int8), t[cut + int(rng.integers(0, 4)):]])
            if len(t) == 0:
                t = q[:1].copy()
            tlen = len(t)
        else:
            t = rng.integers(0, 25, tlen).astype(np.int8)
        if it % 5 == 0:
            q[rng.integers(0, qlen)] |= -128      # SEED_MASK bit must be ignored (basic/value.h:62)
        # any band intersecting the matrix: diagonals range over [-(tlen-1), qlen-1]
        d0 = int(rng.integers(-(tlen - 1) - 5, qlen + 3))
        d1 = d0 + int(rng.integers(1, 140 if it % 7 else 300))
        if d1 <= -(tlen - 1) or d0 >= qlen:
            continue
        cbs = rng.integers(-3, 2, qlen).astype(np.int8) if it % 3 else None
        _compare(q, cbs, t, d0, d1, M, 11, 1, force_p=(2 if it % 11 == 0 else 0))


STAT_KEYS = "score q_begin q_end s_begin s_end length identities mismatches gap_openings gaps".split()


def test_emulated_stats_passes_match_oracle_and_reference():
    """Statistics-without-traceback path (ForwardCell + reversed BackwardCell pass)."""
    hdr, recs = read_tap(os.path.join(GOLDEN, "swipe_long.tap"))
    n = 0
    for rec in recs:
        if rec["hsp_values"] == 0:
            continue
        for t in rec["targets"][:2]:
            rc, o = orc.swipe_stats(rec["query"], rec["cbs"], t["seq"], t["d_begin"], t["d_end"], hdr["matrix8"], 11, 1, 510)
            rc2, e = emu.swipe_stats(rec["query"], rec["cbs"], t["seq"], t["d_begin"], t["d_end"], hdr["matrix8"], 11, 1)
            assert rc == 0 and rc2 == 0
            for k in STAT_KEYS:
                assert e[k] == o[k], (k, e, o)
            n += 1
    assert n >= 2
    hdr, recs = read_tap(os.path.join(GOLDEN, "swipe_fast.tap"), max_records=60)
    for rec in recs[::5]:
        for t in rec["targets"]:
            rc, o = orc.swipe_stats(rec["query"], rec["cbs"], t["seq"], t["d_begin"], t["d_end"], hdr["matrix8"], 11, 1, 510)
            rc2, e = emu.swipe_stats(rec["query"], rec["cbs"], t["seq"], t["d_begin"], t["d_end"], hdr["matrix8"], 11, 1)
            for k in STAT_KEYS:
                assert e[k] == o[k], (k, e, o)


def test_emulated_wavefront_wide_classes_and_ties():
    """Every band class the register-window lanes are built for (P = 1, 2, 4 with one end-cell record per diagonal; P = 8 with
    one per lane), on score ties: repeats of one short motif give many cells with the best score, so the end cell is decided
    by the reference's tie rule (smallest column, then largest row) across diagonals, lanes and step parities."""
    hdr, _ = read_tap(os.path.join(GOLDEN, "swipe_fast.tap"), max_records=1)
    M = hdr["matrix8"]
    rng = np.random.default_rng(12)
    for it in range(120):
        motif = rng.integers(0, 20, int(rng.integers(2, 6))).astype(np.int8)
        q = np.resize(motif, int(rng.integers(5, 120)))
        t = np.resize(motif, int(rng.integers(5, 160)))
        if it % 3 == 0:                                   # break the repeat here and there: several disjoint equal-score segments
            q = q.copy(); t = t.copy()
            q[rng.integers(0, len(q), 2)] = 23
            t[rng.integers(0, len(t), 3)] = 23
        d0 = int(rng.integers(-(len(t) - 1), len(q) - 1))
        d1 = d0 + int(rng.integers(1, 200))
        if d1 <= -(len(t) - 1) or d0 >= len(q):
            continue
        cbs = rng.integers(-2, 2, len(q)).astype(np.int8) if it % 2 else None
        _compare(q, cbs, t, d0, d1, M, 11, 1, force_p=(1, 2, 4, 8)[it % 4])
