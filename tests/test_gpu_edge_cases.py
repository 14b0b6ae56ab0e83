"""-m gpu: edge cases of the hot path through the C ABI -- ragged and degenerate inputs the reference accepts (sequences
shorter than a seed, all-X sequences, a single sequence, no hits at all, a query identical to a target, very long
sequences against short ones) and the error behaviour of misuse (no blocks, no limits, oversized bands)."""
import numpy as np
import pytest
import torch

import oracle_py as orc
from diamond_amd import hip, workload
from test_oracle_seed import blosum62_matrix8, hit_multiset

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    assert torch.cuda.is_available()
    c = hip.Context()
    yield c
    c.close()


def _block(seqs):
    lens = np.array([len(s) for s in seqs], np.int64)
    off = np.concatenate([[0], np.cumsum(lens)])
    data = np.concatenate(seqs).astype(np.int8) if len(seqs) else np.zeros(0, np.int8)
    return workload.sequence_set(data, off)


def _oracle_hits(sp, qd, ql, td, tl, m8):
    cfg = dict(seedp_bits=sp.seedp_bits, index_chunks=sp.index_chunks, hamming_filter_id=sp.hamming_filter_id,
               shapes=[dict(length=sp.shape_len[i], weight=sp.shape_weight[i], mask=sp.shape_mask[i],
                            positions=[sp.shape_pos[i][k] for k in range(sp.shape_weight[i])]) for i in range(sp.n_shapes)],
               reduction=[sp.reduction[i] for i in range(32)], seed_complexity_cut=sp.seed_complexity_cut,
               ungapped_evalue=10000.0 if sp.use_ungapped else 0.0, query_contexts=1)
    return orc.seed_search(orc.seed_cfg_from_tap(cfg, m8), qd, ql, td, tl)


@pytest.mark.parametrize("mode", ["fast", "default"])
def test_ragged_and_degenerate_sequences(ctx, mode):
    rng = np.random.default_rng(1)
    core = rng.integers(0, 20, 400).astype(np.int8)
    queries = [core[:5],                                   # shorter than any seed
               core[:16], core[:17],                       # exactly one / two windows of the 16-long fast shape
               np.full(60, 23, np.int8),                   # all X: no seed at all
               core.copy(),                                # identical to a target
               np.concatenate([core[:200], np.full(30, 23, np.int8), core[230:]]),      # X run in the middle
               rng.integers(0, 20, 3000).astype(np.int8)]  # long, unrelated
    targets = [core.copy(), core[100:300].copy(), core[:12], np.full(40, 23, np.int8), rng.integers(0, 20, 5000).astype(np.int8),
               np.concatenate([rng.integers(0, 20, 700).astype(np.int8), core, rng.integers(0, 20, 900).astype(np.int8)])]
    qd, ql = _block(queries)
    td, tl = _block(targets)
    params = hip.default_params()
    sp, gf = hip.seed_params_preset(mode, params, threads=1)
    ctx.upload_block(hip.QUERY, qd, ql)
    ctx.upload_block(hip.TARGET, td, tl)
    ctx.set_db_letters(float(sum(len(t) for t in targets)))
    ctx.set_gapped_filter(gf)
    ctx.set_query_contexts(1)
    hits = ctx.seed_search(sp)
    want = _oracle_hits(sp, qd, ql, td, tl, blosum62_matrix8())
    assert hit_multiset(hits) == hit_multiset(want) and len(hits) > 10
    m, _ = ctx.extend(qd, td, hits, threads=2)
    by_q = {}
    for r in m:
        by_q.setdefault(int(r["query"]), []).append(r)
    assert set(by_q) >= {4, 5} and not ({0, 3} & set(by_q))
    best = by_q[4][0]                                       # the identical pair: full-length, all identities
    assert (best["target"], best["hsp"]["length"], best["hsp"]["identities"], best["hsp"]["gaps"]) == (0, 400, 400, 0)
    assert 5 in {int(r["target"]) for r in by_q[4]}         # the copy embedded in a longer target
    # every reported alignment re-derives on the oracle with the reported band
    M = hip.matrix_of(params)
    for r in m:
        q, t = queries[int(r["query"])], targets[int(r["target"])]
        cbs = hip.extend_plan(params, *_block([q]), *_block([q]), np.zeros(0, hip.SEED_HIT_DTYPE))[0][256:256 + len(q)]
        rc, o, _ = orc.banded_swipe(q, cbs, t, int(r["d_begin"]), int(r["d_end"]), M, 11, 1, orc.TRACEBACK)
        assert rc == 0 and all(o[k] == r["hsp"][k] for k in ("score", "q_begin", "q_end", "s_begin", "s_end", "identities", "length"))


def test_no_hits_single_sequence_and_empty_calls(ctx):
    rng = np.random.default_rng(2)
    qd, ql = _block([rng.integers(0, 20, 120).astype(np.int8)])
    td, tl = _block([rng.integers(0, 20, 90).astype(np.int8)])
    ctx.upload_block(hip.QUERY, qd, ql)
    ctx.upload_block(hip.TARGET, td, tl)
    ctx.set_gapped_filter(0.0)
    hits = ctx.seed_search(hip.seed_params_fast(1))
    assert len(hits) == 0
    m, _ = ctx.extend(qd, td, hits)
    assert len(m) == 0
    out, _ = ctx.banded_swipe(np.zeros(0, hip.DP_TARGET_DTYPE), hip.SWIPE_SCORE)
    assert len(out) == 0
    ctx.set_gapped_filter(1.0)
    assert len(ctx.gapped_filter(np.zeros(0, hip.SEED_HIT_DTYPE))) == 0
    ctx.set_gapped_filter(0.0)


def test_misuse_is_reported_not_crashed():
    c = hip.Context()
    try:
        with pytest.raises(hip.DiamondHipError):
            c.seed_search(hip.seed_params_fast(1))                               # no blocks
        data = np.full(600, 31, np.int8)
        c.upload_block(hip.QUERY, data)                                          # no limits
        c.upload_block(hip.TARGET, data)
        with pytest.raises(hip.DiamondHipError):
            c.seed_search(hip.seed_params_fast(1))
        with pytest.raises(hip.DiamondHipError):
            c.mask_block(hip.QUERY)
        it = np.zeros(1, hip.DP_TARGET_DTYPE)
        it["query_off"], it["target_off"], it["cbs_off"], it["query_len"], it["target_len"] = 256, 256, -1, 50, 50
        it["d_begin"], it["d_end"] = -40000, 40000                               # band wider than DMND_MAX_BAND (65536)
        with pytest.raises(hip.DiamondHipError, match="Band size"):
            c.banded_swipe(it, hip.SWIPE_SCORE)
        it["d_begin"], it["d_end"] = 10, 10                                      # empty band
        with pytest.raises(hip.DiamondHipError):
            c.banded_swipe(it, hip.SWIPE_SCORE)
        it["d_begin"], it["d_end"], it["query_len"] = -10, 10, 5000              # item outside the uploaded block
        with pytest.raises(hip.DiamondHipError):
            c.banded_swipe(it, hip.SWIPE_SCORE)
        with pytest.raises(hip.DiamondHipError):
            c.set_query_contexts(3)
    finally:
        c.close()


def test_extension_without_kept_traces_gives_the_same_records(monkeypatch):
    """Round 1 normally sweeps in traceback mode and keeps its trace rows (dmnd_swipe_keep). When the rows of a call do not fit
    the context's trace budget the call falls back to a score-only sweep and round 2 sweeps again: same records either way
    (the budget is read from DMND_TRACE_ARENA_MB when the context is created; 16 MB is less than this batch needs)."""
    from diamond_amd import synth, workload
    db, doff, q, qoff = synth.generate(300, members=10, queries=1500, seed=4)
    qd, ql = workload.sequence_set(q, qoff)
    td, tl = workload.sequence_set(db, doff)
    params = hip.default_params()
    params.db_letters = float(doff[-1])
    out = []
    for mb in (None, "16"):
        if mb is None:
            monkeypatch.delenv("DMND_TRACE_ARENA_MB", raising=False)
        else:
            monkeypatch.setenv("DMND_TRACE_ARENA_MB", mb)
        ctx = hip.Context(params=params)
        try:
            ctx.upload_block(hip.QUERY, qd, ql)
            ctx.upload_block(hip.TARGET, td, tl)
            hits = ctx.seed_search(hip.seed_params_fast(threads=4))
            m, _ = ctx.extend(qd, td, hits, threads=4)
            st = ctx.extend_stats()
            torch.cuda.synchronize()
            ctx.touch_streams()                                # what bench.py does after the device-wide synchronize of its timing contract
            m_again_same_ctx, _ = ctx.extend(qd, td, hits, threads=4)
            assert np.array_equal(m, m_again_same_ctx)
        finally:
            ctx.close()
        out.append((m.copy(), st))
    (m_keep, st_keep), (m_again, st_again) = out
    assert len(m_keep) > 500 and len(m_keep) == len(m_again), (len(m_keep), len(m_again))
    for name in m_keep.dtype.names:
        if name == "hsp":
            for f in m_keep["hsp"].dtype.names:
                bad = np.nonzero(m_keep["hsp"][f] != m_again["hsp"][f])[0]
                assert bad.size == 0, ("hsp." + f, bad[:5], m_keep["hsp"][f][bad[:5]], m_again["hsp"][f][bad[:5]])
        else:
            bad = np.nonzero(m_keep[name] != m_again[name])[0]
            assert bad.size == 0, (name, bad[:5], m_keep[name][bad[:5]], m_again[name][bad[:5]])
    assert st_keep["round2_swipe_kernel_ms"] == 0.0 and st_again["round2_swipe_kernel_ms"] > 0.0      # the two paths really differ
    assert st_keep["round1_targets"] * 15_000 > 16 << 20                                                 # ... because the trace records (half a byte per cell) exceed 16 MB
