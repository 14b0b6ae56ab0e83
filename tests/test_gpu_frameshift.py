"""-m gpu parity tests of the three-frame banded sweep kernels (frameshift alignment, blastx -F) through the C ABI
(dmnd_frameshift_swipe): the reference's own calls (tests/golden/f3_*.tap, tapped at banded_3frame_swipe, src/dp/dp.h:296) --
score-only with its 16-channel vector batches, traceback with read coordinates, statistics and transcripts incl. the frameshift
operations -- and random items against the oracle."""
import os
import numpy as np
import pytest
import torch

import oracle_py as orc
from tapfile import read_3frame_tap
from diamond_amd import hip
from test_frameshift import KEYS, _random_case

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MAP = dict(qs_begin="read_begin", qs_end="read_end")


@pytest.fixture(scope="module")
def ctx():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    c = hip.Context()
    yield c
    c.close()


def _pack(cases):
    """cases: [(frames, strand, dna_len, group, [(target, d_begin, d_end, cols)])] -> query block, target block, items"""
    q, t, rows = [], [], []
    qo = to = 0
    for frames, strand, dna_len, group, targets in cases:
        offs = []
        for f in frames:
            offs.append(qo)
            q.append(np.asarray(f, np.int8))
            q.append(np.array([31], np.int8))
            qo += len(f) + 1
        for seq, d0, d1, cols in targets:
            rows.append((offs, to, [len(f) for f in frames], len(seq), d0, d1, cols, strand, dna_len, group))
            t.append(np.asarray(seq, np.int8))
            t.append(np.array([31], np.int8))
            to += len(seq) + 1
    return np.concatenate(q), np.concatenate(t), np.array(rows, dtype=hip.FS_TARGET_DTYPE)


@pytest.mark.parametrize("tap", ["f3_k3.tap", "f3_k1.tap"])
def test_golden_reference_calls(ctx, tap):
    hdr, recs = read_3frame_tap(os.path.join(GOLDEN, tap))
    p = hip.default_params()
    p.db_letters = hdr["db_letters"]
    for score_only in (1, 0):
        sel = [r for r in recs if r["score_only"] == score_only]
        cases = [(r["frames"], r["strand"], r["dna_len"], g, [(t["seq"], t["d_begin"], t["d_end"], t["cols"]) for t in r["targets"]]) for g, r in enumerate(sel)]
        qb, tb, items = _pack(cases)
        ctx.upload_block(hip.QUERY, qb)
        ctx.upload_block(hip.TARGET, tb)
        out, tr = ctx.frameshift_swipe(items, score_only, hdr["frame_shift"])
        k = n = 0
        for r in sel:
            by_target = {}
            for h in r["hsps"]:
                by_target.setdefault(h["swipe_target"], []).append(h)
            for t in r["targets"]:
                o = out[k]
                k += 1
                ev = ctx.lib.dmnd_evalue_p(p, int(o["score"]), len(r["frames"][0]), len(t["seq"])) if o["score"] > 0 else 1e9
                if ev > hdr["max_evalue"]:
                    continue
                keys = ("score", "frame", "q_begin", "q_end", "qs_begin", "qs_end") if score_only else KEYS
                match = [h for h in by_target.get(t["target_idx"], []) if all(h[x] == o[MAP.get(x, x)] for x in keys)]
                assert match, (t["target_idx"], o)
                if not score_only:
                    mine = tr[o["transcript_off"]: o["transcript_off"] + o["transcript_len"]]
                    assert any(np.array_equal(h["transcript"][:-1], mine) for h in match)
                    assert tr[o["transcript_off"] + o["transcript_len"]] == 0
                n += 1
        assert k == len(items) and n > 20


def test_random_items_against_oracle(ctx):
    rng = np.random.default_rng(77)
    M = hip.matrix_of(ctx.params)
    cases, meta = [], []
    for it in range(900):
        frames, dna_len, t, d0, d1 = _random_case(rng, M)
        if len(frames[0]) == 0:
            continue
        cases.append((frames, it % 2, dna_len, it // 5, [(t, d0, d1, int(rng.integers(-50, 900)))]))
        meta.append((frames, it % 2, dna_len, t, d0, d1))
    qb, tb, items = _pack(cases)
    ctx.upload_block(hip.QUERY, qb)
    ctx.upload_block(hip.TARGET, tb)
    for fs in (15, 4):
        out, tr = ctx.frameshift_swipe(items, 0, fs)
        n_hit = 0
        for k, (frames, strand, dna_len, t, d0, d1) in enumerate(meta):
            rc, o, otr = orc.frameshift_traceback(frames, strand, dna_len, t, d0, d1, M, 11, 1, fs)
            assert rc == 0 and out[k]["score"] == o["score"], k
            if o["score"] > 0:
                assert all(out[k][MAP.get(x, x)] == o[x] for x in KEYS), (k, out[k], o)
                assert np.array_equal(tr[out[k]["transcript_off"]: out[k]["transcript_off"] + out[k]["transcript_len"]], otr)
                n_hit += 1
        assert n_hit > 300
    # score only: groups of five items in a channel width of 4 -> batches of 4 + 1 per group, each on its batch geometry
    out, _ = ctx.frameshift_swipe(items, 1, 15, channels=4)
    k = 0
    while k < len(items):
        g = [x for x in range(k, len(items)) if items[x]["group"] == items[k]["group"]]
        targets = [dict(d_begin=int(items[x]["d_begin"]), d_end=int(items[x]["d_end"]), cols=int(items[x]["cols"])) for x in g]
        for batch in orc.frameshift_batches(targets, channels=4):
            for j, band, i0, i1, pos0 in batch:
                frames, strand, dna_len, t, d0, d1 = meta[g[j]]
                s, mc, ov = orc.frameshift_score(frames, t, band, i0, i1, pos0, M, 11, 1, 15)
                rg = orc.frameshift_score_range(strand, dna_len, len(frames[0]), band, i0, pos0, mc)
                o = out[g[j]]
                assert (o["score"], o["max_col"], o["q_begin"], o["q_end"], o["read_begin"], o["read_end"], o["frame"]) == (s, mc, rg["q_begin"], rg["q_end"], rg["qs_begin"], rg["qs_end"], rg["frame"]), g[j]
        k = g[-1] + 1
