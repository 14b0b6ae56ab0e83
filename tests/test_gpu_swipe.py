"""-m gpu parity tests of the HIP banded Smith-Waterman, called through the C ABI
(dmnd_banded_swipe / dmnd_banded_swipe_host in include/diamond_hip.h):
  * against the committed golden vectors minted from the genuine reference (tests/golden/*.tap),
  * against the CPU oracle on seeded random inputs incl. the edge geometries the reference's
    band logic allows (bands leaving the matrix, 1-letter sequences, wide bands -> every P class),
  * at BASELINE-config scale through size-independent properties (mode agreement, transcript
    re-scoring, batch-order independence)."""
import os
import numpy as np
import pytest
import torch

import oracle_py as orc
from tapfile import read_tap
from diamond_amd import hip, synth
from gpu_util import pack_records, pack_records_m

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KEYS = "score q_begin q_end s_begin s_end length identities mismatches positives gap_openings gaps".split()


@pytest.fixture(scope="module")
def ctx():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    c = hip.Context()
    yield c
    c.close()


def _transcript(tr, h):
    return tr[h["transcript_off"]: h["transcript_off"] + h["transcript_len"]]


@pytest.mark.parametrize("tap", ["swipe_default.tap", "swipe_fast.tap", "swipe_blastx.tap"])
def test_golden_reference_calls(ctx, tap):
    hdr, recs = read_tap(os.path.join(GOLDEN, tap))
    for v, mode in ((0, hip.SWIPE_SCORE), (510, hip.SWIPE_TRACEBACK)):
        sel = [r for r in recs if r["hsp_values"] == v]
        if not sel:
            continue
        qb, tb, cbs, items, meta = pack_records(sel)
        ctx.upload_block(hip.QUERY, qb)
        ctx.upload_block(hip.TARGET, tb)
        ctx.upload_cbs(cbs)
        out, tr = ctx.banded_swipe(items, mode, v)
        p = hip.default_params()
        p.db_letters = hdr["db_letters"]
        n_checked = 0
        for k, (rec, t) in enumerate(meta):
            hs = [h for h in rec["hsps"] if (h["swipe_target"], h["d_begin"], h["d_end"]) == (t["target_idx"], t["d_begin"], t["d_end"])]
            o = out[k]
            if not hs:
                ev = ctx.lib.dmnd_evalue_p(p, int(o["score"]), len(rec["query"]), t["true_target_len"]) if o["score"] > 0 else 1e9
                assert o["score"] <= 0 or ev > hdr["max_evalue"]
                continue
            h = hs[0]
            assert o["score"] == h["score"]
            if mode == hip.SWIPE_TRACEBACK and h["swipe_bin"] < 3:
                for key in KEYS:
                    assert o[key] == h[key], (key, k)
                assert np.array_equal(_transcript(tr, o), h["transcript"][:-1])
                assert tr[o["transcript_off"] + o["transcript_len"]] == 0
            n_checked += 1
        assert n_checked > 0


@pytest.mark.parametrize("tap", ["swipe_cbs3.tap", "swipe_cbs4.tap"])
def test_golden_reference_calls_with_adjusted_matrices(ctx, tap):
    """--comp-based-stats 3 / 4 (row f4): DpTargets with a composition-adjusted matrix of their own, as the reference swept them
    (tests/golden/make_cbs_golden.sh). Both call shapes: resident blocks + dmnd_upload_matrices, and the reference's own
    (dmnd_banded_swipe_host with DpTarget::matrix pointers)."""
    hdr, recs = read_tap(os.path.join(GOLDEN, tap))
    p = hip.default_params()
    p.db_letters = hdr["db_letters"]
    n_adj = 0
    for v, mode in ((0, hip.SWIPE_SCORE), (510, hip.SWIPE_TRACEBACK)):
        sel = [r for r in recs if r["hsp_values"] == v]
        assert sel
        qb, tb, cbs, items, meta, mats = pack_records_m(sel)
        assert len(mats) > 20 and (items["cbs_off"] <= -2).sum() == len(mats)
        ctx.upload_block(hip.QUERY, qb)
        ctx.upload_block(hip.TARGET, tb)
        ctx.upload_cbs(cbs)
        ctx.upload_matrices(mats)
        out, tr = ctx.banded_swipe(items, mode, v)
        for k, (rec, t) in enumerate(meta):
            hs = [h for h in rec["hsps"] if (h["swipe_target"], h["d_begin"], h["d_end"]) == (t["target_idx"], t["d_begin"], t["d_end"])]
            o = out[k]
            if not hs:
                ev = ctx.lib.dmnd_evalue_p(p, int(o["score"]), len(rec["query"]), t["true_target_len"]) if o["score"] > 0 else 1e9
                assert o["score"] <= 0 or ev > hdr["max_evalue"]
                continue
            h = hs[0]
            assert o["score"] == h["score"], (k, t["matrix"] is not None)
            if mode == hip.SWIPE_TRACEBACK and h["swipe_bin"] < 3:
                for key in KEYS:
                    assert o[key] == h[key], (key, k)
                assert np.array_equal(_transcript(tr, o), h["transcript"][:-1])
            n_adj += t["matrix"] is not None
        # the reference's call shape, one query at a time
        for rec in sel[:25]:
            targets = [(t["seq"], t["d_begin"], t["d_end"], t["matrix"]) for t in rec["targets"]]
            ho, htr = ctx.banded_swipe_host(rec["query"], rec["cbs"], targets, mode, v)
            for j, t in enumerate(rec["targets"]):
                hs = [h for h in rec["hsps"] if (h["swipe_target"], h["d_begin"], h["d_end"]) == (t["target_idx"], t["d_begin"], t["d_end"])]
                if hs:
                    assert ho[j]["score"] == hs[0]["score"]
                    if mode == hip.SWIPE_TRACEBACK and hs[0]["swipe_bin"] < 3:
                        assert np.array_equal(_transcript(htr, ho[j]), hs[0]["transcript"][:-1])
    assert n_adj > 50
    ctx.upload_matrices(np.zeros((0, 32, 32), np.int8))


@pytest.mark.parametrize("rows", ["0", "1"])
def test_random_matrices_against_oracle(ctx, rows, monkeypatch):
    """Every item with a random matrix of its own (or the context's, one in four), every band class up to the multi-wavefront
    sweep, all modes incl. the statistics cells: equal to the oracle run on that matrix. rows = "1": the items on the context's
    matrix take the row classes of the packed sweeps where their bands fit, the others (32-bit kernels) never do."""
    monkeypatch.setenv("DMND_SWEEP_ROWS", rows)
    M = hip.matrix_of(ctx.params)
    rng = np.random.default_rng(2024)
    recs = _random_items(rng, 240, M) + _random_items(rng, 60, M, wide=True)
    for k, rec in enumerate(recs):
        t = rec["targets"][0]
        if k % 4:
            m = M.copy()
            m[:24, :24] = np.clip(m[:24, :24].astype(int) + rng.integers(-2, 3, (24, 24)), -128, 127).astype(np.int8)
            t["matrix"] = m
            rec["cbs_used"] = None
        else:
            t["matrix"] = None
            rec["cbs_used"] = rec["cbs"]
    qb, tb, cbs, items, meta, mats = pack_records_m(recs)
    ctx.upload_block(hip.QUERY, qb)
    ctx.upload_block(hip.TARGET, tb)
    ctx.upload_cbs(cbs)
    ctx.upload_matrices(mats)
    res = {mode: ctx.banded_swipe(items, mode) for mode in (hip.SWIPE_SCORE, hip.SWIPE_COORDS, hip.SWIPE_TRACEBACK)}
    st, _ = ctx.banded_swipe(items, hip.SWIPE_STATS, 510)
    for k, (rec, t) in enumerate(meta):
        m = t["matrix"] if t["matrix"] is not None else M
        rc, o, otr = orc.banded_swipe(rec["query"], rec["cbs_used"], t["seq"], t["d_begin"], t["d_end"], m, 11, 1, orc.TRACEBACK)
        assert rc == 0
        assert res[hip.SWIPE_SCORE][0][k]["score"] == o["score"], k
        assert res[hip.SWIPE_COORDS][0][k]["score"] == o["score"]
        if o["score"] > 0:
            g, tr = res[hip.SWIPE_TRACEBACK][0][k], res[hip.SWIPE_TRACEBACK][1]
            for key in KEYS:
                assert g[key] == o[key], (key, k)
            assert np.array_equal(_transcript(tr, g), otr)
            rc, so = orc.swipe_stats(rec["query"], rec["cbs_used"], t["seq"], t["d_begin"], t["d_end"], m, 11, 1, 510)
            assert rc == 0
            for key in "score q_begin q_end s_begin s_end length identities mismatches gap_openings gaps".split():
                assert st[k][key] == so[key], (key, k)
    # an item that names a matrix that was not uploaded is refused
    bad = items[:1].copy()
    bad["cbs_off"] = -2 - len(mats)
    with pytest.raises(hip.DiamondHipError):
        ctx.banded_swipe(bad, hip.SWIPE_SCORE)
    ctx.upload_matrices(np.zeros((0, 32, 32), np.int8))


def _random_items(rng, n, M, wide=False):
    recs = []
    for it in range(n):
        qlen = int(rng.integers(1, 400 if not wide else 1500))
        q = rng.integers(0, 25, qlen).astype(np.int8)
        if it % 2 == 0:
            t = q.copy()
            mut = rng.random(qlen) < 0.35
            t[mut] = rng.integers(0, 20, int(mut.sum()))
            cut = int(rng.integers(0, qlen))
            t = np.concatenate([t[:cut], rng.integers(0, 20, int(rng.integers(0, 8))).astype(np.int8), t[cut + int(rng.integers(0, 5)):]])
            if len(t) == 0:
                t = q[:1].copy()
        else:
            t = rng.integers(0, 25, int(rng.integers(1, 400))).astype(np.int8)
        tlen = len(t)
        if it % 5 == 0:
            q[rng.integers(0, qlen)] |= -128
        d0 = int(rng.integers(-(tlen - 1) - 5, qlen + 3))
        width = int(rng.integers(1, 140)) if not wide else int(rng.integers(100, 1100))
        d1 = d0 + width
        if d1 <= -(tlen - 1) or d0 >= qlen:
            d0, d1 = -3, 4
        cbs = rng.integers(-3, 2, qlen).astype(np.int8) if it % 3 else None
        recs.append({"query": q, "cbs": cbs, "targets": [{"seq": t, "d_begin": d0, "d_end": d1}]})
    return recs


@pytest.mark.parametrize("wide,rows", [(False, "0"), (False, "1"), (True, "1")])
def test_random_geometry_against_oracle(ctx, wide, rows, monkeypatch):
    monkeypatch.setenv("DMND_SWEEP_ROWS", rows)      # the row classes (bands of <= 96 / 160 diagonals) off / on whatever the item count
    M = hip.matrix_of(ctx.params)
    rng = np.random.default_rng(11 + wide)
    recs = _random_items(rng, 400 if not wide else 120, M, wide)
    qb, tb, cbs, items, meta = pack_records(recs)
    ctx.upload_block(hip.QUERY, qb)
    ctx.upload_block(hip.TARGET, tb)
    ctx.upload_cbs(cbs)
    res = {}
    for mode in (hip.SWIPE_SCORE, hip.SWIPE_COORDS, hip.SWIPE_TRACEBACK):
        res[mode] = ctx.banded_swipe(items, mode)
    for k, (rec, t) in enumerate(meta):
        rc, o, otr = orc.banded_swipe(rec["query"], rec["cbs"], t["seq"], t["d_begin"], t["d_end"], M, 11, 1, orc.TRACEBACK)
        assert rc == 0
        assert res[hip.SWIPE_SCORE][0][k]["score"] == o["score"]
        c = res[hip.SWIPE_COORDS][0][k]
        assert c["score"] == o["score"]
        if o["score"] > 0:
            assert (c["q_end"], c["s_end"]) == (o["q_end"], o["s_end"])
            g, tr = res[hip.SWIPE_TRACEBACK][0][k], res[hip.SWIPE_TRACEBACK][1]
            for key in KEYS:
                assert g[key] == o[key], (key, k)
            assert np.array_equal(_transcript(tr, g), otr)


@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 17, 150])
def test_row_classes_take_any_number_of_items(ctx, n, monkeypatch):
    """Bands of up to 96 / 160 diagonals are swept by the row classes (eight items per wavefront, one pair per 16-lane DPP row): any
    item count (wavefronts filled up with copies that store nothing), items of very different lengths in one wavefront, and the
    same numbers as the wavefront classes (DMND_SWEEP_ROWS=0) and the oracle."""
    M = hip.matrix_of(ctx.params)
    rng = np.random.default_rng(100 + n)
    recs = _random_items(rng, n, M)
    for k, r in enumerate(recs):                                       # one class per call: 3 for odd n, 5 for even n
        t = r["targets"][0]
        width = int(rng.integers(1, 97)) if n % 2 else int(rng.integers(129, 161))
        t["d_begin"] = -min(len(t["seq"]) - 1, width // 2 + int(rng.integers(0, 5)))
        t["d_end"] = t["d_begin"] + width
        if k == 1:                                                     # a long item next to short ones
            q = rng.integers(0, 20, 2500).astype(np.int8)
            r["query"], r["cbs"] = q, None
            t["seq"] = q.copy()
    qb, tb, cbs, items, meta = pack_records(recs)
    ctx.upload_block(hip.QUERY, qb)
    ctx.upload_block(hip.TARGET, tb)
    ctx.upload_cbs(cbs)
    monkeypatch.setenv("DMND_SWEEP_ROWS", "1")        # (by default only calls of 131 072 items and more take the row classes)
    res = {mode: ctx.banded_swipe(items, mode) for mode in (hip.SWIPE_SCORE, hip.SWIPE_COORDS, hip.SWIPE_TRACEBACK)}
    monkeypatch.setenv("DMND_SWEEP_ROWS", "0")
    off = {mode: ctx.banded_swipe(items, mode) for mode in (hip.SWIPE_SCORE, hip.SWIPE_COORDS, hip.SWIPE_TRACEBACK)}
    monkeypatch.delenv("DMND_SWEEP_ROWS")
    for mode in res:
        assert np.array_equal(res[mode][0], off[mode][0]) and np.array_equal(res[mode][1], off[mode][1])
    for k, (rec, t) in enumerate(meta):
        rc, o, otr = orc.banded_swipe(rec["query"], rec["cbs"], t["seq"], t["d_begin"], t["d_end"], M, 11, 1, orc.TRACEBACK)
        assert rc == 0
        assert res[hip.SWIPE_SCORE][0][k]["score"] == o["score"]
        c = res[hip.SWIPE_COORDS][0][k]
        assert c["score"] == o["score"]
        if o["score"] > 0:
            assert (c["q_end"], c["s_end"]) == (o["q_end"], o["s_end"])
            g, tr = res[hip.SWIPE_TRACEBACK][0][k], res[hip.SWIPE_TRACEBACK][1]
            for key in KEYS:
                assert g[key] == o[key], (key, k)
            assert np.array_equal(_transcript(tr, g), otr)


def test_host_call_shape_equals_batched_call(ctx):
    """dmnd_banded_swipe_host is the literal DP::BandedSwipe::swipe call shape (one query + its targets)."""
    hdr, recs = read_tap(os.path.join(GOLDEN, "swipe_fast.tap"), max_records=40)
    M = hip.matrix_of(ctx.params)
    for rec in recs[:40:4]:
        mode = hip.SWIPE_SCORE if rec["hsp_values"] == 0 else hip.SWIPE_TRACEBACK
        out, tr = ctx.banded_swipe_host(rec["query"], rec["cbs"], [(t["seq"], t["d_begin"], t["d_end"]) for t in rec["targets"]], mode)
        for o, t in zip(out, rec["targets"]):
            rc, ref, rtr = orc.banded_swipe(rec["query"], rec["cbs"], t["seq"], t["d_begin"], t["d_end"], M, 11, 1,
                                            orc.SCORE_ONLY if mode == hip.SWIPE_SCORE else orc.TRACEBACK)
            assert o["score"] == ref["score"]
            if mode == hip.SWIPE_TRACEBACK and ref["score"] > 0:
                assert np.array_equal(_transcript(tr, o), rtr)


def test_trace_arena_chunking_is_transparent(ctx):
    """Small trace arena -> many chunks; results must not change."""
    M = hip.matrix_of(ctx.params)
    recs = _random_items(np.random.default_rng(3), 300, M)
    qb, tb, cbs, items, _ = pack_records(recs)
    ctx.upload_block(hip.QUERY, qb)
    ctx.upload_block(hip.TARGET, tb)
    ctx.upload_cbs(cbs)
    a, atr = ctx.banded_swipe(items, hip.SWIPE_TRACEBACK)
    os.environ["DMND_TRACE_ARENA_MB"] = "64"
    c2 = hip.Context()
    del os.environ["DMND_TRACE_ARENA_MB"]
    c2.upload_block(hip.QUERY, qb)
    c2.upload_block(hip.TARGET, tb)
    c2.upload_cbs(cbs)
    b, btr = c2.banded_swipe(items, hip.SWIPE_TRACEBACK)
    c2.close()
    assert np.array_equal(a, b) and np.array_equal(atr, btr)


def _rescore(q, cbs, t, h, tr, M):
    """Re-scores a packed transcript (PackedOperation codes) -> (score, q_end, s_end)."""
    i, j, s = int(h["q_begin"]), int(h["s_begin"]), 0
    k = 0
    ops = tr[h["transcript_off"]: h["transcript_off"] + h["transcript_len"]]
    while k < len(ops):
        op, cnt = int(ops[k]) >> 6, int(ops[k]) & 63
        if op == 0 or op == 3:
            n = cnt if op == 0 else 1
            for _ in range(n):
                s += int(M[t[j] & 31, q[i] & 31]) + (int(cbs[i]) if cbs is not None else 0)
                i += 1
                j += 1
            k += 1
        elif op == 1:
            run = 0
            while k < len(ops) and int(ops[k]) >> 6 == 1:
                run += int(ops[k]) & 63
                k += 1
            s -= 11 + run
            i += run
        else:
            run = 0
            while k < len(ops) and int(ops[k]) >> 6 == 2:
                run += 1
                k += 1
            s -= 11 + run
            j += run
    return s, i, j


@pytest.mark.parametrize("rows", [None, "1"])
def test_baseline_scale_properties(ctx, rows, monkeypatch):
    """C1-shaped synthetic workload (1k queries x 10k-seq DB, SURVEY.md 8d generator), ~20k DpTargets:
    size-independent invariants instead of a per-item oracle. rows = "1": in the row classes of the packed sweeps (20 000 items
    are below the count from which they are taken by default)."""
    if rows:
        monkeypatch.setenv("DMND_SWEEP_ROWS", rows)
    db, doff, q, qoff = synth.generate(1000, members=10, queries=1000, seed=1)
    rng = np.random.default_rng(2)
    # every query against 20 members: its own family is unknown here, so pair with random targets plus
    # a self-derived band around the main diagonal; half the items get narrow bands, half wide
    nq = len(qoff) - 1
    qi = np.repeat(np.arange(nq), 20)
    ti = rng.integers(0, len(doff) - 1, qi.size)
    ql = (qoff[qi + 1] - qoff[qi]).astype(np.int32)
    tl = (doff[ti + 1] - doff[ti]).astype(np.int32)
    centre = rng.integers(-20, 20, qi.size)
    half = rng.choice([12, 30, 40, 64, 150], qi.size)
    items = np.zeros(qi.size, dtype=hip.DP_TARGET_DTYPE)
    items["query_off"] = qoff[qi]
    items["target_off"] = doff[ti]
    items["cbs_off"] = -1
    items["query_len"] = ql
    items["target_len"] = tl
    items["d_begin"] = np.maximum(centre - half, -(tl - 1))
    items["d_end"] = np.minimum(centre + half + 1, ql)
    ctx.upload_block(hip.QUERY, q)
    ctx.upload_block(hip.TARGET, db)
    ctx.upload_cbs(np.zeros(0, np.int8))
    s0, _ = ctx.banded_swipe(items, hip.SWIPE_SCORE)
    s1, _ = ctx.banded_swipe(items, hip.SWIPE_COORDS)
    s2, tr = ctx.banded_swipe(items, hip.SWIPE_TRACEBACK)
    assert np.array_equal(s0["score"], s1["score"]) and np.array_equal(s0["score"], s2["score"])
    pos = s0["score"] > 0
    assert pos.sum() > 1000
    assert np.array_equal(s1["q_end"][pos], s2["q_end"][pos]) and np.array_equal(s1["s_end"][pos], s2["s_end"][pos])
    # batch-order independence
    perm = rng.permutation(items.size)
    sp, _ = ctx.banded_swipe(items[perm], hip.SWIPE_SCORE)
    assert np.array_equal(sp["score"], s0["score"][perm])
    # transcripts re-score to the reported score and coordinates; spot-check against the oracle
    M = hip.matrix_of(ctx.params)
    for k in rng.choice(np.nonzero(pos)[0], 300, replace=False):
        it = items[k]
        qs = q[it["query_off"]: it["query_off"] + it["query_len"]]
        ts = db[it["target_off"]: it["target_off"] + it["target_len"]]
        sc, qe, se = _rescore(qs, None, ts, s2[k], tr, M)
        assert (sc, qe, se) == (s2[k]["score"], s2[k]["q_end"], s2[k]["s_end"])
        assert s2[k]["length"] == s2[k]["identities"] + s2[k]["mismatches"] + s2[k]["gaps"]
    for k in rng.choice(items.size, 200, replace=False):
        it = items[k]
        qs = q[it["query_off"]: it["query_off"] + it["query_len"]]
        ts = db[it["target_off"]: it["target_off"] + it["target_len"]]
        rc, o, _ = orc.banded_swipe(qs, None, ts, it["d_begin"], it["d_end"], M, 11, 1, orc.SCORE_ONLY)
        assert o["score"] == s0[k]["score"]


STAT_KEYS = "score q_begin q_end s_begin s_end length identities mismatches gap_openings gaps".split()


def test_stats_without_traceback_golden_and_oracle(ctx):
    """DMND_SWIPE_STATS = ForwardCell pass + reversed BackwardCell pass (swipe_wrapper.cpp:364-444): the long-protein
    golden calls (DP size > max_swipe_dp) and a seeded sample against the oracle."""
    hdr, recs = read_tap(os.path.join(GOLDEN, "swipe_long.tap"))
    sel = [r for r in recs if r["hsp_values"] != 0]
    qb, tb, cbs, items, meta = pack_records(sel)
    ctx.upload_block(hip.QUERY, qb)
    ctx.upload_block(hip.TARGET, tb)
    ctx.upload_cbs(cbs)
    out, _ = ctx.banded_swipe(items, hip.SWIPE_STATS, 510)
    n = 0
    for k, (rec, t) in enumerate(meta):
        hs = [h for h in rec["hsps"] if (h["swipe_target"], h["d_begin"], h["d_end"]) == (t["target_idx"], t["d_begin"], t["d_end"])]
        if hs and hs[0]["swipe_bin"] >= 3:
            for key in STAT_KEYS:
                assert out[k][key] == hs[0][key], (key, k)
            n += 1
    assert n >= 5
    M = hip.matrix_of(ctx.params)
    recs = _random_items(np.random.default_rng(21), 300, M)
    qb, tb, cbs, items, meta = pack_records(recs)
    ctx.upload_block(hip.QUERY, qb)
    ctx.upload_block(hip.TARGET, tb)
    ctx.upload_cbs(cbs)
    out, _ = ctx.banded_swipe(items, hip.SWIPE_STATS, 510)
    for k, (rec, t) in enumerate(meta):
        rc, o = orc.swipe_stats(rec["query"], rec["cbs"], t["seq"], t["d_begin"], t["d_end"], M, 11, 1, 510)
        assert rc == 0 and out[k]["score"] == o["score"]
        if o["score"] > 0:          # a zero score never becomes an HSP; its coordinates are don't-care
            for key in STAT_KEYS:
                assert out[k][key] == o[key], (key, k, out[k], o)


@pytest.mark.parametrize("rows", ["0", "1"])
def test_saturated_items_are_rerun_in_32_bits(ctx, rows, monkeypatch):
    """Scores of 32767 and above saturate the packed 16-bit sweep: such items must come back from the 32-bit kernels with the
    oracle's numbers in every mode, and must not disturb the items that shared their wavefront (rows = "1": the row classes, where
    the re-run is in another class than the first sweep)."""
    monkeypatch.setenv("DMND_SWEEP_ROWS", rows)
    M = hip.matrix_of(ctx.params)
    rng = np.random.default_rng(77)
    recs = _random_items(rng, 9, M)
    big = np.full(3400, 17, np.int8)                     # W x W = 11 per column: 37400
    recs.insert(3, {"query": big, "cbs": None, "targets": [{"seq": big.copy(), "d_begin": -20, "d_end": 21}]})
    mid = rng.integers(0, 20, 3100).astype(np.int8)      # below and above the limit in one batch
    recs.insert(6, {"query": mid, "cbs": None, "targets": [{"seq": mid.copy(), "d_begin": -30, "d_end": 31}]})
    qb, tb, cbs, items, meta = pack_records(recs)
    ctx.upload_block(hip.QUERY, qb)
    ctx.upload_block(hip.TARGET, tb)
    ctx.upload_cbs(cbs)
    res = {mode: ctx.banded_swipe(items, mode) for mode in (hip.SWIPE_SCORE, hip.SWIPE_COORDS, hip.SWIPE_TRACEBACK)}
    seen_big = False
    for k, (rec, t) in enumerate(meta):
        rc, o, otr = orc.banded_swipe(rec["query"], rec["cbs"], t["seq"], t["d_begin"], t["d_end"], M, 11, 1, orc.TRACEBACK)
        assert rc == 0
        seen_big |= o["score"] >= 32767
        assert res[hip.SWIPE_SCORE][0][k]["score"] == o["score"]
        c = res[hip.SWIPE_COORDS][0][k]
        assert c["score"] == o["score"]
        if o["score"] > 0:
            assert (c["q_end"], c["s_end"]) == (o["q_end"], o["s_end"])
            g, tr = res[hip.SWIPE_TRACEBACK][0][k], res[hip.SWIPE_TRACEBACK][1]
            for key in KEYS:
                assert g[key] == o[key], (key, k)
            assert np.array_equal(_transcript(tr, g), otr)
    assert seen_big


def test_bands_wider_than_one_wavefront(ctx):
    """Merged bands of long repeat proteins: above 4096 diagonals (2048 with statistics) an item is swept by several wavefronts
    of one workgroup. Tandem-repeat pairs with bands from 2049 to 20000 in every mode against the oracle; they share the batch
    with ordinary items, which must not be disturbed (round 1 failed the whole batch on the first over-wide band)."""
    M = hip.matrix_of(ctx.params)
    rng = np.random.default_rng(77)
    recs = _random_items(rng, 24, M)
    unit = rng.integers(0, 20, 37).astype(np.int8)
    for width, qlen, tlen in ((2049, 2600, 2400), (3000, 1800, 3500), (4096, 4200, 4100), (4097, 4300, 4200), (6000, 5000, 3300),
                              (9000, 5200, 5100), (20000, 10500, 10200)):
        q = np.tile(unit, qlen // 37 + 1)[:qlen].copy()
        t = np.tile(unit, tlen // 37 + 1)[:tlen].copy()
        for a in (q, t):                                        # diverged copies of the repeat, a few indels
            mut = rng.random(len(a)) < 0.2
            a[mut] = rng.integers(0, 20, int(mut.sum()))
        t = np.concatenate([t[:500], t[517:2000], rng.integers(0, 20, 9).astype(np.int8), t[2000:]])
        d0 = int(rng.integers(-(len(t) - 1), qlen - width)) if qlen + len(t) - 1 > width else -(len(t) - 1)
        d0 = max(-(len(t) - 1) - 3, min(d0, -width // 2))       # around the main diagonal
        cbs = rng.integers(-2, 2, qlen).astype(np.int8)
        recs.append({"query": q, "cbs": cbs, "targets": [{"seq": t, "d_begin": d0, "d_end": d0 + width}]})
    qb, tb, cbs, items, meta = pack_records(recs)
    ctx.upload_block(hip.QUERY, qb)
    ctx.upload_block(hip.TARGET, tb)
    ctx.upload_cbs(cbs)
    res = {mode: ctx.banded_swipe(items, mode) for mode in (hip.SWIPE_SCORE, hip.SWIPE_COORDS, hip.SWIPE_TRACEBACK)}
    stats, _ = ctx.banded_swipe(items, hip.SWIPE_STATS, 510)
    n_wide = 0
    for k, (rec, t) in enumerate(meta):
        rc, o, otr = orc.banded_swipe(rec["query"], rec["cbs"], t["seq"], t["d_begin"], t["d_end"], M, 11, 1, orc.TRACEBACK)
        assert rc == 0
        assert res[hip.SWIPE_SCORE][0][k]["score"] == o["score"], k
        c = res[hip.SWIPE_COORDS][0][k]
        assert c["score"] == o["score"]
        if o["score"] > 0:
            assert (c["q_end"], c["s_end"]) == (o["q_end"], o["s_end"])
            g, tr = res[hip.SWIPE_TRACEBACK][0][k], res[hip.SWIPE_TRACEBACK][1]
            for key in KEYS:
                assert g[key] == o[key], (key, k)
            assert np.array_equal(_transcript(tr, g), otr)
            rc, so = orc.swipe_stats(rec["query"], rec["cbs"], t["seq"], t["d_begin"], t["d_end"], M, 11, 1, 510)
            assert rc == 0
            for key in STAT_KEYS:
                assert stats[k][key] == so[key], (key, k)
        if t["d_end"] - t["d_begin"] > 2048:
            n_wide += 1
            assert o["score"] > 500                              # the repeat pair really aligns
    assert n_wide == 7
