"""-m gpu parity tests of the HIP seed stage (dmnd_seed_search, include/diamond_hip.h), called through the C ABI:
hit multiset equality with the stage-2 hits tapped from the genuine reference at Extension::extend (golden), with the
oracle on other partitionings and on a C1-sized synthetic workload, plus size-independent properties."""
import os
import numpy as np
import pytest
import torch

import oracle_py as orc
import emu_py as emu
from tapfile import read_ext_tap
from diamond_amd import hip, synth
from test_oracle_seed import hit_multiset

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def hit_set(h):
    return set(zip(h["query"].tolist(), h["subject"].tolist(), h["seed_offset"].tolist(), h["score"].tolist()))


@pytest.fixture(scope="module")
def ctx():
    assert torch.cuda.is_available()
    c = hip.Context()
    yield c
    c.close()


def to_hip_params(cfg):
    e = emu.seed_params_from_tap(cfg)
    p = hip.SeedParams()
    import ctypes
    assert ctypes.sizeof(p) == ctypes.sizeof(e)
    ctypes.memmove(ctypes.byref(p), ctypes.byref(e), ctypes.sizeof(p))
    return p


@pytest.mark.parametrize("tap", ["ext_fast.tap", "ext_fast_synth.tap", "ext_6x10.tap", "ext_default.tap", "ext_default_synth.tap", "ext_sensitive.tap", "ext_blastx.tap", "ext_bjz.tap"])
def test_seed_hits_equal_reference(ctx, tap):
    # ext_bjz.tap: query windows that share a seed while one of them holds B / J / Z at a care position (see test_oracle_seed.py)
    cfg, recs = read_ext_tap(os.path.join(GOLDEN, tap))
    ctx.upload_block(hip.QUERY, cfg["query"]["data"], cfg["query"]["limits"])
    ctx.upload_block(hip.TARGET, cfg["target"]["data"], cfg["target"]["limits"])
    hits = ctx.seed_search(to_hip_params(cfg))
    ref = np.concatenate([r["hits"] for r in recs])
    assert len(hits) == len(ref) and hit_multiset(hits) == hit_multiset(ref)
    # sorted by query as the extension stage needs them
    assert (np.diff(hits["query"].astype(np.int64)) >= 0).all()


@pytest.mark.parametrize("tap", ["ext_hashed.tap", "ext_hashed_default.tap", "ext_hashed_sens.tap"])
def test_query_indexed_seed_hits_equal_reference(ctx, tap):
    """--algo 1 (the reference's query-indexed algorithm, its AUTO choice at BASELINE's C2-C4 sizes): goldens minted with
    --algo 1 on sequences with masked runs (also at sequence starts), stop codons and ambiguity letters."""
    cfg, recs = read_ext_tap(os.path.join(GOLDEN, tap))
    ctx.upload_block(hip.QUERY, cfg["query"]["data"], cfg["query"]["limits"])
    ctx.upload_block(hip.TARGET, cfg["target"]["data"], cfg["target"]["limits"])
    hits = ctx.seed_search(to_hip_params(dict(cfg, seed_encoding=1)))
    ref = np.concatenate([r["hits"] for r in recs])
    assert len(hits) == len(ref) and hit_multiset(hits) == hit_multiset(ref)


def test_buffer_overflow_retry_paths_are_transparent(ctx, monkeypatch):
    """Joined-position and hit buffers that are too small (forced through the test hooks) must grow and give the same hits;
    with several shapes the shapes after the overflowing one run with zero remaining capacity."""
    cfg, recs = read_ext_tap(os.path.join(GOLDEN, "ext_6x10.tap"))
    ctx.upload_block(hip.QUERY, cfg["query"]["data"], cfg["query"]["limits"])
    ctx.upload_block(hip.TARGET, cfg["target"]["data"], cfg["target"]["limits"])
    want = ctx.seed_search(to_hip_params(cfg))
    for mcap, hcap in ((1, None), (5000, None), (None, 10), (3000, 100)):
        if mcap is not None:
            monkeypatch.setenv("DMND_SEED_MATCHED_CAP", str(mcap))
        if hcap is not None:
            monkeypatch.setenv("DMND_SEED_HIT_CAP", str(hcap))
        got = ctx.seed_search(to_hip_params(cfg))
        monkeypatch.delenv("DMND_SEED_MATCHED_CAP", raising=False)
        monkeypatch.delenv("DMND_SEED_HIT_CAP", raising=False)
        assert np.array_equal(got, want)
    # default sensitivity: the buffer of deferred pairs (stage-2 scores above 255, resolved in a second pass) overflows too
    cfg, recs = read_ext_tap(os.path.join(GOLDEN, "ext_default.tap"))
    ctx.upload_block(hip.QUERY, cfg["query"]["data"], cfg["query"]["limits"])
    ctx.upload_block(hip.TARGET, cfg["target"]["data"], cfg["target"]["limits"])
    want = ctx.seed_search(to_hip_params(cfg))
    assert (want["score"] > 255).any() and (want["score"] == 255).any()
    monkeypatch.setenv("DMND_SEED_DEFERRED_CAP", "7")
    monkeypatch.setenv("DMND_SEED_HIT_CAP", "50")
    got = ctx.seed_search(to_hip_params(cfg))
    monkeypatch.delenv("DMND_SEED_DEFERRED_CAP")
    monkeypatch.delenv("DMND_SEED_HIT_CAP")
    assert np.array_equal(got, want)


@pytest.mark.parametrize("tap,tiled", [("ext_default.tap", "0"), ("ext_default.tap", "1"), ("ext_sensitive.tap", "0")])
def test_folded_need_map_of_the_deferred_pass_equals_reference(ctx, tap, tiled, monkeypatch):
    """Round 5: the deferred pass gathers the joined positions of the seeds that have a deferred pair through a folded copy of the
    need map kept in LDS (taken by itself from 4 M joined positions per shape: C3) -- forced on the goldens that have deferred
    pairs, long seeds (plain and tiled pair filter) and short: the hits equal the plain gather's and the reference's."""
    cfg, recs = read_ext_tap(os.path.join(GOLDEN, tap))
    ctx.upload_block(hip.QUERY, cfg["query"]["data"], cfg["query"]["limits"])
    ctx.upload_block(hip.TARGET, cfg["target"]["data"], cfg["target"]["limits"])
    monkeypatch.setenv("DMND_SEED_TILED", tiled)
    monkeypatch.setenv("DMND_SEED_COLLECT_FOLDED_FROM", str(1 << 60))
    plain = ctx.seed_search(to_hip_params(cfg))
    monkeypatch.setenv("DMND_SEED_COLLECT_FOLDED_FROM", "1")
    hits = ctx.seed_search(to_hip_params(cfg))
    monkeypatch.delenv("DMND_SEED_COLLECT_FOLDED_FROM")
    monkeypatch.delenv("DMND_SEED_TILED")
    assert (plain["score"] >= 255).any()                       # the deferred pass ran
    ref = np.concatenate([r["hits"] for r in recs])
    assert len(hits) == len(ref) and hit_multiset(hits) == hit_multiset(ref)
    assert np.array_equal(hits, plain)


@pytest.mark.parametrize("tap", ["ext_fast_synth.tap", "ext_default.tap", "ext_sensitive.tap", "ext_blastx.tap", "ext_rank.tap"])
def test_tiled_pair_filter_equals_reference(ctx, tap, monkeypatch):
    """The LDS-tiled pair filter (joined positions sorted by seed; normally chosen for >= 4 M joined positions per shape)
    forced on the goldens: same hit multiset as the reference."""
    cfg, recs = read_ext_tap(os.path.join(GOLDEN, tap))
    ctx.upload_block(hip.QUERY, cfg["query"]["data"], cfg["query"]["limits"])
    ctx.upload_block(hip.TARGET, cfg["target"]["data"], cfg["target"]["limits"])
    monkeypatch.setenv("DMND_SEED_TILED", "1")
    hits = ctx.seed_search(to_hip_params(cfg))
    monkeypatch.setenv("DMND_SEED_SURVIVOR_CAP", "5")          # survivor list too small: grows and reruns
    assert np.array_equal(ctx.seed_search(to_hip_params(cfg)), hits)
    monkeypatch.delenv("DMND_SEED_SURVIVOR_CAP")
    monkeypatch.setenv("DMND_SEED_TILED", "0")
    plain = ctx.seed_search(to_hip_params(cfg))
    monkeypatch.delenv("DMND_SEED_TILED")
    ref = np.concatenate([r["hits"] for r in recs])
    assert len(hits) == len(ref) and hit_multiset(hits) == hit_multiset(ref)
    assert np.array_equal(hits, plain)


@pytest.mark.parametrize("tap,enc", [("ext_fast_synth.tap", 0), ("ext_default.tap", 0), ("ext_6x10.tap", 0), ("ext_blastx.tap", 0), ("ext_rank.tap", 0), ("ext_bjz.tap", 0),
                                     ("ext_hashed.tap", 1), ("ext_hashed_default.tap", 1)])
def test_by_class_stream_for_long_seeds_equals_reference(ctx, tap, enc, monkeypatch):
    """Round 5: the reference stream by key class -- eight workgroups per tile group, each probing its XCD's eighth of the level-1
    filter and of the table -- for LONG seeds (the path a query block above 2^24 positions takes by itself: C5), forced on the goldens:
    the hit multiset is the reference's, and the hits equal the plain stream's."""
    cfg, recs = read_ext_tap(os.path.join(GOLDEN, tap))
    ctx.upload_block(hip.QUERY, cfg["query"]["data"], cfg["query"]["limits"])
    ctx.upload_block(hip.TARGET, cfg["target"]["data"], cfg["target"]["limits"])
    params = to_hip_params(dict(cfg, seed_encoding=1) if enc else cfg)
    monkeypatch.setenv("DMND_SEED_CLASSES_LONG", "0")
    plain = ctx.seed_search(params)
    monkeypatch.setenv("DMND_SEED_CLASSES_LONG", "1")
    hits = ctx.seed_search(params)
    monkeypatch.delenv("DMND_SEED_CLASSES_LONG")
    ref = np.concatenate([r["hits"] for r in recs])
    assert len(hits) == len(ref) and hit_multiset(hits) == hit_multiset(ref)
    assert np.array_equal(hits, plain)


@pytest.mark.parametrize("tap,enc", [("ext_sensitive.tap", 0), ("ext_hashed_sens.tap", 1)])
def test_short_seed_buffers_grow_without_changing_the_hits(ctx, tap, enc, monkeypatch):
    """The fused short-seed pass with joined-position / survivor buffers that start far too small (they overflow, the shape is run
    again with larger ones): same hits as the reference and as the run that never overflows."""
    cfg, recs = read_ext_tap(os.path.join(GOLDEN, tap))
    ctx.upload_block(hip.QUERY, cfg["query"]["data"], cfg["query"]["limits"])
    ctx.upload_block(hip.TARGET, cfg["target"]["data"], cfg["target"]["limits"])
    params = to_hip_params(dict(cfg, seed_encoding=1) if enc else cfg)
    plain = ctx.seed_search(params)
    monkeypatch.setenv("DMND_SEED_MATCHED_CAP", "700")
    monkeypatch.setenv("DMND_SEED_SURVIVOR_CAP", "40")
    assert np.array_equal(ctx.seed_search(params), plain)
    monkeypatch.delenv("DMND_SEED_MATCHED_CAP")
    monkeypatch.delenv("DMND_SEED_SURVIVOR_CAP")
    ref = np.concatenate([r["hits"] for r in recs])
    assert len(plain) == len(ref) and hit_multiset(plain) == hit_multiset(ref)


@pytest.mark.parametrize("chunks,bits", [(1, 8), (3, 9), (7, 10)])
def test_seed_hits_equal_oracle_other_partitionings(ctx, chunks, bits):
    cfg, _ = read_ext_tap(os.path.join(GOLDEN, "ext_fast_synth.tap"), max_records=1)
    cfg = dict(cfg, index_chunks=chunks, seedp_bits=bits)
    ctx.upload_block(hip.QUERY, cfg["query"]["data"], cfg["query"]["limits"])
    ctx.upload_block(hip.TARGET, cfg["target"]["data"], cfg["target"]["limits"])
    hits = ctx.seed_search(to_hip_params(cfg))
    a = orc.seed_search(orc.seed_cfg_from_tap(cfg), cfg["query"]["data"], cfg["query"]["limits"], cfg["target"]["data"], cfg["target"]["limits"])
    assert len(hits) == len(a) > 300 and hit_set(hits) == hit_set(a)


def _blocks(data, off):
    """SequenceSet layout (data/string_set.h:27-60): 256 x 0x1F, then seq, 0x1F, seq, 0x1F, ..., 256 x 0x1F."""
    n = len(off) - 1
    lens = np.diff(off)
    limits = 256 + np.concatenate([[0], np.cumsum(lens + 1)])
    out = np.full(int(limits[-1]) + 256, 31, np.int8)
    idx = np.repeat(limits[:-1] - off[:-1], lens) + np.arange(off[-1])
    out[idx] = data
    return out, limits.astype(np.int64)


def test_seed_stage_c1_scale_against_oracle_and_properties(ctx):
    """BASELINE config C1 shape (1k queries x 10k sequences): full oracle comparison + invariants."""
    db, doff, q, qoff = synth.generate(1000, members=10, queries=1000, seed=1)
    qd, ql = _blocks(q, qoff)
    td, tl = _blocks(db, doff)
    p = hip.seed_params_fast(threads=8)
    ctx.upload_block(hip.QUERY, qd, ql)
    ctx.upload_block(hip.TARGET, td, tl)
    hits = ctx.seed_search(p)
    import ctypes
    oc = orc.SeedCfg()
    oc.seedp_bits, oc.index_chunks, oc.hamming_filter_id, oc.n_shapes = p.seedp_bits, p.index_chunks, p.hamming_filter_id, p.n_shapes
    oc.shape_len[0], oc.shape_weight[0], oc.shape_mask[0] = p.shape_len[0], p.shape_weight[0], p.shape_mask[0]
    for k in range(p.shape_weight[0]):
        oc.shape_pos[0][k] = p.shape_pos[0][k]
    for i in range(32):
        oc.reduction[i] = p.reduction[i]
    oc.reduction_size, oc.ungapped_window, oc.left_most_interval, oc.seed_complexity_cut = 10, 48, 32, p.seed_complexity_cut
    oc.tile_size, oc.simd_lanes = p.tile_size, p.simd_lanes
    a = orc.seed_search(oc, qd, ql, td, tl)
    assert len(hits) == len(a) > 1000 and hit_set(hits) == hit_set(a)
    # properties: every hit is a true seed match inside its sequences, deterministic across runs
    again = ctx.seed_search(p)
    assert np.array_equal(hits, again)
    pos = np.array([p.shape_pos[0][k] for k in range(p.shape_weight[0])])
    red = np.array([p.reduction[i] for i in range(32)])
    qloc = ql[hits["query"]] + hits["seed_offset"]
    assert (red[qd[qloc[:, None] + pos[None, :]] & 31] == red[td[hits["subject"][:, None] + pos[None, :]] & 31]).all()
    assert (qloc + 16 <= ql[hits["query"] + 1] - 1).all()


@pytest.mark.parametrize("tap", ["ext_sensitive.tap", "ext_hashed_sens.tap"])
def test_fused_short_seed_pipeline_equals_the_list_based_one_and_survives_small_buffers(ctx, tap, monkeypatch):
    """Short seeds (weight < 10) take the per-shape pipeline with the Hamming filter fused into the reference stream. Same sorted
    hit list as the list-based path (DMND_SEED_FUSED=0) and as the reference's hits; joined-position, survivor and hit buffers that
    start too small (test hooks) grow -- the hit buffer keeping the hits of the earlier shapes."""
    cfg, recs = read_ext_tap(os.path.join(GOLDEN, tap))
    enc = 1 if "hashed" in tap else 0
    ctx.upload_block(hip.QUERY, cfg["query"]["data"], cfg["query"]["limits"])
    ctx.upload_block(hip.TARGET, cfg["target"]["data"], cfg["target"]["limits"])
    p = to_hip_params(dict(cfg, seed_encoding=enc))
    fused = ctx.seed_search(p)
    ref = np.concatenate([r["hits"] for r in recs])
    assert len(fused) == len(ref) and hit_multiset(fused) == hit_multiset(ref)
    monkeypatch.setenv("DMND_SEED_FUSED", "0")
    assert np.array_equal(ctx.seed_search(p), fused)
    monkeypatch.delenv("DMND_SEED_FUSED")
    for env in ({"DMND_SEED_MATCHED_CAP": "7"}, {"DMND_SEED_SURVIVOR_CAP": "3"}, {"DMND_SEED_HIT_CAP": "5"},
                {"DMND_SEED_MATCHED_CAP": "100", "DMND_SEED_SURVIVOR_CAP": "10", "DMND_SEED_HIT_CAP": "40"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = ctx.seed_search(p)
        for k in env:
            monkeypatch.delenv(k)
        assert np.array_equal(got, fused), env


@pytest.mark.parametrize("mode,hashed", [("fast", 0), ("default", 0), ("sensitive", 0), ("sensitive", 1), ("fast", 1)])
def test_query_index_reuse_over_reference_blocks(mode, hashed):
    """One query block against three reference blocks with dmnd_set_query_index_reuse: the kept index gives the same hits as a
    fresh context per block pair; re-uploading or masking the query block, or other seed parameters, rebuild it."""
    from diamond_amd import synth
    db, doff, q, qoff = synth.generate(300, members=6, queries=200, seed=77)
    rng = np.random.default_rng(5)
    for a, off in ((db, doff), (q, qoff)):                         # low-complexity stretches: masked seeds, erased groups
        for i in range(0, len(off) - 1, 9):
            b, e = int(off[i]), int(off[i + 1])
            if e - b > 60:
                p = int(rng.integers(b, e - 30))
                a[p:p + 24] = np.tile(np.array([int(rng.integers(0, 20)), int(rng.integers(0, 20))], np.int8), 12)
    qd, ql = _blocks(q, qoff)
    cuts = [0, 500, 1100, len(doff) - 1]
    tblocks = [_blocks(db[doff[a]:doff[b]], doff[a:b + 1] - doff[a]) for a, b in zip(cuts, cuts[1:])]
    params = hip.default_params()
    sp, _ = hip.seed_params_preset(mode, params, threads=4)
    if hashed:
        hip.set_query_indexed(sp, threads=4)

    def fresh(td, tl, qdata=qd):
        c = hip.Context(params=params)
        try:
            c.upload_block(hip.QUERY, qdata, ql)
            c.upload_block(hip.TARGET, td, tl)
            return c.seed_search(sp)
        finally:
            c.close()

    c = hip.Context(params=params)
    try:
        c.set_query_index_reuse(True)
        c.upload_block(hip.QUERY, qd, ql)
        n_hits = 0
        for rep in range(2):                                        # second round: every call finds the index ready
            for td, tl in tblocks:
                c.upload_block(hip.TARGET, td, tl)
                got = c.seed_search(sp)
                assert np.array_equal(got, fresh(td, tl))
                n_hits += len(got)
        assert n_hits > 200
        # the query block changes: the kept index must not be used
        q2 = qd.copy()
        q2[ql[3]:ql[3] + 40] = 23
        c.upload_block(hip.QUERY, q2, ql)
        c.upload_block(hip.TARGET, *tblocks[0])
        assert np.array_equal(c.seed_search(sp), fresh(*tblocks[0], qdata=q2))
        # other parameters
        sp2, _ = hip.seed_params_preset("default" if mode != "default" else "fast", params, threads=4)
        c2 = hip.Context(params=params)
        c2.upload_block(hip.QUERY, q2, ql)
        c2.upload_block(hip.TARGET, *tblocks[0])
        want2 = c2.seed_search(sp2)
        c2.close()
        assert np.array_equal(c.seed_search(sp2), want2)
        assert np.array_equal(c.seed_search(sp), fresh(*tblocks[0], qdata=q2))
    finally:
        c.close()
