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


def _expand_runs(blocks, rng):
    """--max-hsps: every record becomes a run of 1-4 records of its (query, target) pair; hsp.q_begin = position inside the run."""
    multi = []
    for b in blocks:
        rows = []
        for r in b:
            n = int(rng.integers(1, 5))
            g = np.repeat(r[None], n)
            g["hsp"]["score"][1:] = np.sort(rng.integers(10, int(r["hsp"]["score"]) + 1, n - 1))[::-1]
            g["evalue"][1:] = 1.0                                  # worse than any first record: must not re-rank the run
            g["hsp"]["q_begin"] = np.arange(n)
            rows.append(g)
        multi.append(np.concatenate(rows) if rows else np.zeros(0, hip.MATCH_DTYPE))
    return multi


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


def test_hsp_records_of_a_target_move_through_the_join_together():
    """--max-hsps: a match is a run of consecutive records of one (query, target) pair, the first one carrying the target's rank
    (Match::filter_evalue); the join orders the runs by their first records, keeps the order inside a run and counts TARGETS
    against -k (the heap rule JoinRecord::same_subject_, output/join_blocks.cpp:129-137,180-206)."""
    rng = np.random.default_rng(13)
    blocks = _block_records(rng, 50, 4, 400)
    multi = _expand_runs(blocks, rng)
    for k in (25, 2):
        got = hip.join_blocks(np.concatenate(multi[::-1]), k)
        want = hip.join_blocks(np.concatenate(blocks[::-1]), k)      # the same join on the first records alone
        heads = got[got["hsp"]["q_begin"] == 0]
        assert [(int(r["query"]), int(r["target"])) for r in heads] == [(int(r["query"]), int(r["target"])) for r in want]
        runs = {}
        for i, r in enumerate(got):
            runs.setdefault((int(r["query"]), int(r["target"])), []).append((i, int(r["hsp"]["q_begin"])))
        src = {}
        for b in multi:
            for r in b:
                src[(int(r["query"]), int(r["target"]))] = src.get((int(r["query"]), int(r["target"])), 0) + 1
        for key, rs in runs.items():
            assert [x[1] for x in rs] == list(range(src[key])) and rs[-1][0] - rs[0][0] == len(rs) - 1, key      # complete, in order, consecutive
    top = np.concatenate(multi)
    top["bit_score"] = 0.39 * top["hsp"]["score"] + 3.1
    got = hip.join_blocks_top(top, 10.0)
    assert (got["hsp"]["q_begin"] == 0).sum() < (top["hsp"]["q_begin"] == 0).sum() and set(np.diff(np.flatnonzero(np.r_[got["hsp"]["q_begin"] == 0, True]))) <= {1, 2, 3, 4}


def test_global_ranking_table_update():
    """dmnd_rank_update = merge_hits (align/global_ranking/table.cpp:135-151): per target the best score seen in any block, the row
    in (score descending, ordinal ascending) order, cut at n; score 0 = empty entry."""
    rng = np.random.default_rng(14)
    nq, n = 30, 6
    table = np.zeros(nq * n, hip.RANKED_DTYPE)
    best = [dict() for _ in range(nq)]
    for block in range(5):
        rows = []
        for q in range(nq):
            if rng.random() < 0.2:
                continue
            for t in rng.choice(40, int(rng.integers(1, 12)), replace=False):
                sc = int(rng.integers(1, 20))                        # few distinct values: ties broken by the ordinal
                rows.append((q, int(t), sc, int(rng.integers(0, 6)), 0))
                if sc > best[q].get(int(t), (0, 0))[0]:
                    best[q][int(t)] = (sc, rows[-1][3])
        hip.rank_update(table, n, np.array(rows, hip.RANKED_DTYPE))
        # what the table must hold now: the n best of everything seen so far -- as long as nothing that fell out of the table comes
        # back with a lower score than it had (the reference forgets dropped targets too; the test data avoids the case by checking
        # only rows whose history fits the table)
        for q in range(nq):
            row = table[q * n:(q + 1) * n]
            filled = row[row["score"] > 0]
            assert all((int(a["score"]), -int(a["target"])) >= (int(b["score"]), -int(b["target"])) for a, b in zip(filled, filled[1:]))
            assert len(set(filled["target"].tolist())) == len(filled)
            if len(best[q]) <= n:
                want = sorted(((-s, t) for t, (s, c) in best[q].items()))      # (the context of two equal scores of one target is either one's)
                assert [(-int(r["score"]), int(r["target"])) for r in filled] == want
