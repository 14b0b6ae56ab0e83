"""Pins the CPU restatement (oracle/banded_swipe.c, oracle/evalue.c) against known answers minted from
the genuine reference at its own dispatch seam (tests/golden/make_swipe_golden.sh ->
DP::BandedSwipe::swipe, /root/reference/src/dp/dp.h:287).  CPU only."""
import os
import numpy as np
import pytest

import oracle_py as orc
from tapfile import read_tap

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TAPS = ["swipe_default.tap", "swipe_fast.tap", "swipe_long.tap", "swipe_blastx.tap"]
COORD_KEYS = "q_begin q_end s_begin s_end length identities mismatches gap_openings gaps".split()


def _targets_with_hsps(rec):
    by_t = {}
    for h in rec["hsps"]:
        by_t.setdefault((h["swipe_target"], h["d_begin"], h["d_end"]), []).append(h)
    for t in rec["targets"]:
        yield t, by_t.get((t["target_idx"], t["d_begin"], t["d_end"]), [])


@pytest.mark.parametrize("tap", TAPS)
def test_restated_swipe_matches_reference(tap):
    hdr, recs = read_tap(os.path.join(GOLDEN, tap))
    M, go, ge = hdr["matrix8"], hdr["gap_open"], hdr["gap_extend"]
    ev = orc.evaluer(hdr["db_letters"], go, ge)
    n_hsp = n_silent = 0
    for rec in recs:
        q, cbs, v = rec["query"], rec["cbs"], rec["hsp_values"]
        for t, hsps in _targets_with_hsps(rec):
            assert orc.banded_cols(len(q), len(t["seq"]), t["d_begin"], t["d_end"]) == t["cols"]
            if not hsps:
                # the reference dropped it: score <= 0 or e-value above the report cutoff (banded_swipe.h:334-336)
                rc, o, _ = orc.banded_swipe(q, cbs, t["seq"], t["d_begin"], t["d_end"], M, go, ge, orc.SCORE_ONLY)
                assert rc == 0
                assert o["score"] <= 0 or orc.evalue(ev, o["score"], len(q), t["true_target_len"]) > hdr["max_evalue"]
                n_silent += 1
                continue
            assert len(hsps) == 1
            h = hsps[0]
            if v == 0:
                rc, o, _ = orc.banded_swipe(q, cbs, t["seq"], t["d_begin"], t["d_end"], M, go, ge, orc.SCORE_ONLY)
                assert rc == 0 and o["score"] == h["score"]
            elif h["swipe_bin"] < 3:        # traceback bins (swipe_wrapper.cpp:191-194)
                rc, o, tr = orc.banded_swipe(q, cbs, t["seq"], t["d_begin"], t["d_end"], M, go, ge, orc.TRACEBACK)
                assert rc == 0 and o["score"] == h["score"]
                for k in COORD_KEYS + ["positives"]:
                    assert o[k] == h[k], (k, o, h)
                assert h["transcript"][-1] == 0 and np.array_equal(h["transcript"][:-1], tr)
            else:                            # statistics without traceback + reversed pass (:364-444)
                rc, o = orc.swipe_stats(q, cbs, t["seq"], t["d_begin"], t["d_end"], M, go, ge, v)
                assert rc == 0 and o["score"] == h["score"]
                for k in COORD_KEYS:
                    assert o[k] == h[k], (k, o, h)
            assert orc.evalue(ev, h["score"], len(q), t["true_target_len"]) == pytest.approx(h["evalue"], rel=1e-12, abs=0)
            assert orc.bitscore(ev, h["score"]) == pytest.approx(h["bit_score"], rel=1e-12)
            n_hsp += 1
    assert n_hsp > 0


@pytest.mark.parametrize("tap", ["swipe_cbs3.tap", "swipe_cbs4.tap"])
def test_restated_swipe_with_adjusted_matrices(tap):
    """--comp-based-stats 3 / 4: a DpTarget that carries a composition-adjusted matrix is swept with it and WITHOUT the query's
    Hauser bias (banded_swipe.h:157-165, swipe.h:43-54: the bias vector is blended to zero in the target's channel)."""
    hdr, recs = read_tap(os.path.join(GOLDEN, tap))
    M, go, ge = hdr["matrix8"], hdr["gap_open"], hdr["gap_extend"]
    ev = orc.evaluer(hdr["db_letters"], go, ge)
    n_adj = n_plain = 0
    for rec in recs:
        q, v = rec["query"], rec["hsp_values"]
        for t, hsps in _targets_with_hsps(rec):
            own = t["matrix"] is not None
            m = M
            if own:
                m = np.full((32, 32), -128, np.int8)
                m[:26] = t["matrix"]
            cbs = None if own else rec["cbs"]
            mode = orc.SCORE_ONLY if v == 0 else orc.TRACEBACK
            rc, o, tr = orc.banded_swipe(q, cbs, t["seq"], t["d_begin"], t["d_end"], m, go, ge, mode)
            assert rc == 0
            if not hsps:
                assert o["score"] <= 0 or orc.evalue(ev, o["score"], len(q), t["true_target_len"]) > hdr["max_evalue"]
                continue
            h = hsps[0]
            assert o["score"] == h["score"]
            if v != 0 and h["swipe_bin"] < 3:
                for k in COORD_KEYS + ["positives"]:
                    assert o[k] == h[k], (k, o, h)
                assert np.array_equal(h["transcript"][:-1], tr)
            n_adj += own
            n_plain += not own
    assert n_adj > 50 and (n_plain > 50 or tap == "swipe_cbs4.tap")


def test_golden_covers_all_modes():
    seen = set()
    for tap in TAPS:
        _, recs = read_tap(os.path.join(GOLDEN, tap))
        for rec in recs:
            for h in rec["hsps"]:
                seen.add((rec["hsp_values"] != 0, h["swipe_bin"] >= 3))
    assert {(False, False), (True, False), (True, True)} <= seen


def test_int8_lane_mask_leak_never_reaches_a_result():
    """SURVEY 8 row a15, the corner DESIGN.md section 2 lists: in the reference's 8-bit vector pass a channel whose band is narrower
    than the vector's band has the lane mask -128 (not minus infinity) ADDED to vgap, hgap and the match scores of the out-of-band
    parts (banded_swipe.h:277-297), so a gap value E >= 128 at a band edge survives as E - 128 in an out-of-band cell. The device
    kernels give every target exactly its own band. That is the same result: the leaked value can only re-enter the band through
    the gap of the cell next to it, where it is at least 128 + gap_open + 2 gap_extend - (largest mismatch penalty) below the diagonal
    predecessor's contribution to that very cell. Here the claim is checked on the reference's own work items: every golden target
    with a score in the range where the 8-bit pass is the final one and gaps of 128 and more exist (139..254) is swept as one
    channel of a wider vector band, mask -128 (oracle_set_channel_band), in several placements -- score, coordinates, statistics
    and transcript stay what the reference reported."""
    rng = np.random.default_rng(5)
    checked = 0
    for tap in TAPS[:2] + ["swipe_fast_synth.tap"]:
        hdr, recs = read_tap(os.path.join(GOLDEN, tap))
        M, go, ge = hdr["matrix8"], hdr["gap_open"], hdr["gap_extend"]
        for rec in recs:
            q, cbs = rec["query"], rec["cbs"]
            for t, hsps in _targets_with_hsps(rec):
                if len(hsps) != 1 or not (128 + go + ge <= hsps[0]["score"] < 255):
                    continue
                h = hsps[0]
                mode = orc.TRACEBACK if (rec["hsp_values"] != 0 and h["swipe_bin"] < 3) else orc.SCORE_ONLY
                for _ in range(3):
                    lo, hi = int(rng.integers(0, 48)), int(rng.integers(0, 48))
                    if lo + hi == 0:
                        lo = 7
                    try:
                        orc.set_channel_band(t["d_begin"], t["d_end"], 128)
                        rc, o, tr = orc.banded_swipe(q, cbs, t["seq"], t["d_begin"] - lo, t["d_end"] + hi, M, go, ge, mode)
                    finally:
                        orc.set_channel_band(0, 0, 0)
                    assert rc == 0 and o["score"] == h["score"], (tap, lo, hi, o["score"], h["score"])
                    if mode == orc.TRACEBACK:
                        for k in COORD_KEYS + ["positives"]:
                            assert o[k] == h[k], (k, lo, hi, o, h)
                        assert np.array_equal(h["transcript"][:-1], tr)
                    checked += 1
    assert checked > 100
