"""-m gpu parity tests of the HIP gapped filter (dmnd_gapped_filter, include/diamond_hip.h) through the
C ABI: per seed hit both filter values against the oracle, and the surviving target sets against the genuine reference
(tap at Extension::gapped_filter, tests/golden/gf_sensitive.tap)."""
import os
import numpy as np
import pytest
import torch

import oracle_py as orc
from tapfile import read_gf_tap, read_ext_tap
from diamond_amd import hip
from test_oracle_seed import blosum62_matrix8

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx():
    assert torch.cuda.is_available()
    c = hip.Context()
    yield c
    c.close()


def test_flags_and_scores_equal_oracle_and_reference(ctx):
    recs = read_gf_tap(os.path.join(GOLDEN, "gf_sensitive.tap"))
    cfg, _ = read_ext_tap(os.path.join(GOLDEN, "ext_sensitive.tap"), max_records=1)
    m8 = blosum62_matrix8()
    qd, ql, td, tl = cfg["query"]["data"], cfg["query"]["limits"], cfg["target"]["data"], cfg["target"]["limits"]
    cbs = np.zeros(int(ql[-1]), np.int8)
    rows, owner = [], []
    for ri, r in enumerate(recs):
        cbs[r["query_offset"]:r["query_offset"] + r["qlen"]] = r["cbs"]
        qid = int(np.searchsorted(ql, r["query_offset"], side="right") - 1)
        for ti, t in enumerate(r["targets"]):
            for hi, hj in t["hits"][:, :2]:
                rows.append((qid, hi, tl[t["block_id"]] + hj, 0, 0))
                owner.append((ri, ti))
    hits = np.array(rows, dtype=hip.SEED_HIT_DTYPE)
    ctx.upload_block(hip.QUERY, qd, ql)
    ctx.upload_block(hip.TARGET, td, tl)
    ctx.upload_cbs(cbs)
    ctx.set_gapped_filter(1.0)
    flags, scores = ctx.gapped_filter(hits, use_cbs=True, with_scores=True)
    assert ctx.gapped_filter_ms() > 0
    survive = {}
    n2 = 0
    for x, (ri, ti) in enumerate(owner):
        r, t = recs[ri], recs[ri]["targets"][ti]
        q = qd[r["query_offset"]:r["query_offset"] + r["qlen"]]
        s = td[tl[t["block_id"]]:tl[t["block_id"] + 1] - 1]
        hi, hj = int(hits["seed_offset"][x]), int(hits["subject"][x] - tl[t["block_id"]])
        f1 = orc.gapped_filter_hit(m8, q, r["cbs"], s, hi, hj, 64, 100, r["diag_score"])
        assert scores[x, 0] == f1
        if f1 > t["cutoff1"]:
            f2 = orc.gapped_filter_hit(m8, q, r["cbs"], s, hi, hj, 128, r["window"], r["diag_score"])
            assert scores[x, 1] == f2 and flags[x] == (f2 > t["cutoff2"])
            n2 += 1
        else:
            assert scores[x, 1] == -1 and flags[x] == 0
        survive[(ri, ti)] = survive.get((ri, ti), False) or bool(flags[x])
    assert len(hits) > 5000 and n2 > 3000
    for (ri, ti), ok in survive.items():                      # the reference's own verdict per (query, target)
        assert ok == (recs[ri]["targets"][ti]["block_id"] in set(recs[ri]["out"].tolist()))
    # without the bias the profile is the plain matrix
    f0, s0 = ctx.gapped_filter(hits[:200], use_cbs=False, with_scores=True)
    for x in range(200):
        ri, ti = owner[x]
        r, t = recs[ri], recs[ri]["targets"][ti]
        q = qd[r["query_offset"]:r["query_offset"] + r["qlen"]]
        s = td[tl[t["block_id"]]:tl[t["block_id"] + 1] - 1]
        assert s0[x, 0] == orc.gapped_filter_hit(m8, q, None, s, int(hits["seed_offset"][x]), int(hits["subject"][x] - tl[t["block_id"]]), 64, 100, 20)


def test_query_length_classes_and_hit_order_against_oracle(ctx, monkeypatch):
    """The kernel builds a query's score profile in LDS for a unit of consecutive hits of that query, in one of three LDS sizes
    by query length (<= 512, <= 1024, <= 1792 letters), and keeps the matrix path for longer queries: queries on both sides of
    every boundary, hits in query order and shuffled (units of one hit), mask letters in both sequences, hits near the sequence
    ends. Both filter values of every hit equal the oracle's; DMND_GF_PROFILE=0 (one wavefront per hit, matrix path) gives the same."""
    from diamond_amd import workload
    rng = np.random.default_rng(3)
    m8 = blosum62_matrix8()
    qlens = [85, 100, 300, 511, 512, 513, 1000, 1024, 1025, 1500, 1792, 1793, 2500]
    qs = [rng.integers(0, 20, n).astype(np.int8) for n in qlens]
    ts = []
    for i, q in enumerate(qs):                                    # targets: mutated pieces of the queries + random flanks
        for _ in range(3):
            a = int(rng.integers(0, max(1, len(q) - 80)))
            piece = q[a:a + int(rng.integers(60, 400))].copy()
            mut = rng.random(len(piece)) < 0.35
            piece[mut] = rng.integers(0, 20, int(mut.sum()))
            t = np.concatenate([rng.integers(0, 20, int(rng.integers(0, 150))).astype(np.int8), piece, rng.integers(0, 20, int(rng.integers(0, 150))).astype(np.int8)])
            t[rng.random(len(t)) < 0.02] = 23                     # X: masked letters
            ts.append(t)
    qs[2][40:60] = 23

    def block(seqs):
        off = np.concatenate([[0], np.cumsum([len(x) for x in seqs])])
        return workload.sequence_set(np.concatenate(seqs).astype(np.int8), off)

    qd, ql = block(qs)
    td, tl = block(ts)
    cbs = rng.integers(-3, 4, int(ql[-1])).astype(np.int8)
    rows = []
    for qi, q in enumerate(qs):
        for ti in range(len(ts)):
            if ti // 3 != qi and rng.random() < 0.8:
                continue
            for _ in range(4):
                rows.append((qi, int(rng.integers(0, len(q))), int(tl[ti] + rng.integers(0, len(ts[ti]))), 0, 0))
        rows.append((qi, 0, int(tl[3 * qi]), 0, 0))
        rows.append((qi, len(q) - 1, int(tl[3 * qi] + len(ts[3 * qi]) - 1), 0, 0))
    hits = np.array(rows, dtype=hip.SEED_HIT_DTYPE)
    ctx.upload_block(hip.QUERY, qd, ql)
    ctx.upload_block(hip.TARGET, td, tl)
    ctx.upload_cbs(cbs)
    ctx.set_gapped_filter(1.0)
    flags, scores = ctx.gapped_filter(hits, use_cbs=True, with_scores=True)
    n2 = 0
    for x in range(0, len(hits), 3):
        qi = int(hits["query"][x])
        ti = int(np.searchsorted(tl, hits["subject"][x], side="right") - 1)
        q, t = qs[qi], ts[ti]
        c = cbs[ql[qi]:ql[qi] + len(q)]
        hi, hj = int(hits["seed_offset"][x]), int(hits["subject"][x] - tl[ti])
        assert scores[x, 0] == orc.gapped_filter_hit(m8, q, c, t, hi, hj, 64, 100, 20), (qi, ti, hi, hj)
        if scores[x, 1] >= 0:
            assert scores[x, 1] == orc.gapped_filter_hit(m8, q, c, t, hi, hj, 128, 200, 20), (qi, ti, hi, hj)
            n2 += 1
    assert len(hits) > 300 and n2 > 30
    perm = rng.permutation(len(hits))
    f2, s2 = ctx.gapped_filter(hits[perm], use_cbs=True, with_scores=True)
    assert np.array_equal(f2, flags[perm]) and np.array_equal(s2, scores[perm])
    monkeypatch.setenv("DMND_GF_PROFILE_TEST", "0")               # read per call (the product's DMND_GF_PROFILE is read once)
    f3, s3 = ctx.gapped_filter(hits, use_cbs=True, with_scores=True)
    assert np.array_equal(f3, flags) and np.array_equal(s3, scores)


def test_filter_off_and_argument_errors(ctx):
    ctx.set_gapped_filter(0.0)
    with pytest.raises(hip.DiamondHipError):
        ctx.gapped_filter(np.zeros(1, dtype=hip.SEED_HIT_DTYPE))
