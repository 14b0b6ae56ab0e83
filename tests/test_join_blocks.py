"""dmnd_join_blocks (host part of the multi-block path) against the restated heap merge of the
reference's join_query (oracle_py.join_blocks): random per-block record lists with ties in e-value and score."""
import numpy as np

import oracle_py as orc
from diamond_amd import hip


def _block_records(rng, n_queries, n_blocks, targets_per_block):
    blocks = []
    for b in range(n_blocks):
        rows = []
        for q in range(n_queries):
            n = int(rng.integers(0, 30))
            if n == 0:
                continue
            t = rng.choice(targets_per_block, n, replace=False) + b * targets_per_block
            score = rng.integers(30, 60, n)                       # few distinct values: ties
            ev = np.round(10.0 ** (-score / 4.0), 12)               # e-value tied whenever the score is
            ev[rng.random(n) < 0.2] = 0.0
            order = sorted(range(n), key=lambda i: (ev[i], -score[i], t[i]))      # a block's own output order (match_less)
            order = order[:25]
            r = np.zeros(len(order), hip.MATCH_DTYPE)
            r["query"], r["target"], r["evalue"] = q, t[order], ev[order]
            r["hsp"]["score"] = score[order]
            r["hsp"]["length"] = rng.integers(1, 1000, len(order))
            rows.append(r)
        blocks.append(np.concatenate(rows) if rows else np.zeros(0, hip.MATCH_DTYPE))
    return blocks


def test_join_blocks_equals_reference_heap_merge():
    rng = np.random.default_rng(11)
    for n_blocks, k in ((1, 25), (2, 25), (5, 25), (7, 3), (4, 100)):
        blocks = _block_records(rng, 60, n_blocks, 500)
        got = hip.join_blocks(np.concatenate(blocks[::-1]), k)          # block order must not matter
        want = []
        for q in range(60):
            per = [[(float(r["evalue"]), int(r["hsp"]["score"]), int(r["target"])) for r in b[b["query"] == q]] for b in blocks]
            want += [(q,) + x for x in orc.join_blocks(per, k)]
        assert [(int(r["query"]), float(r["evalue"]), int(r["hsp"]["score"]), int(r["target"])) for r in got] == want
        # the other fields travel with their record
        src = {(int(r["query"]), int(r["target"])): int(r["hsp"]["length"]) for b in blocks for r in b}
        assert all(src[(int(r["query"]), int(r["target"]))] == int(r["hsp"]["length"]) for r in got)


def test_join_blocks_edge_cases():
    assert len(hip.join_blocks(np.zeros(0, hip.MATCH_DTYPE), 25)) == 0
    r = np.zeros(3, hip.MATCH_DTYPE)
    r["query"] = [2, 0, 2]
    r["target"] = [5, 9, 1]
    r["evalue"] = [1e-5, 1e-3, 1e-5]
    r["hsp"]["score"] = [50, 40, 50]
    out = hip.join_blocks(r, 1)
    assert [(int(x["query"]), int(x["target"])) for x in out] == [(0, 9), (2, 1)]


def test_join_blocks_top_keeps_the_bit_score_window():
    """--top: records of a query over all blocks in (score descending, target ascending) order, kept while
    (1 - bit score / best bit score) * 100 <= percent (GlobalCulling, output/target_culling.h:62-63)."""
    rng = np.random.default_rng(12)
    blocks = _block_records(rng, 40, 4, 300)
    rec = np.concatenate(blocks)
    rec["bit_score"] = 0.39 * rec["hsp"]["score"] + 3.1            # any increasing map of the score
    for pct in (0.0, 5.0, 30.0, 100.0):
        got = hip.join_blocks_top(rec[::-1], pct)
        want = []
        for q in range(40):
            mine = sorted(((-int(r["hsp"]["score"]), int(r["target"]), float(r["bit_score"])) for r in rec[rec["query"] == q]))
            if not mine:
                continue
            top = mine[0][2]
            for negs, t, bits in mine:
                if (1.0 - bits / top) * 100.0 <= pct:
                    want.append((q, t, -negs))
                else:
                    break
        assert [(int(r["query"]), int(r["target"]), int(r["hsp"]["score"])) for r in got] == want, pct
    assert len(hip.join_blocks_top(rec, 100.0)) == len(rec)
