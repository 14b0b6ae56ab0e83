"""Pins the seed-stage restatement (oracle/seed_search.c) against the stage-2 hit lists the
genuine reference hands to Extension::extend (tests/golden/ext_*.tap, minted by make_swipe_golden.sh). CPU only."""
import os
import numpy as np
import pytest

import oracle_py as orc
from tapfile import read_ext_tap

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def hit_set(h):
    return set(zip(h["query"].tolist(), h["subject"].tolist(), h["seed_offset"].tolist(), h["score"].tolist()))


def hit_multiset(h):
    """Sorted tuple list: with many shapes (--sensitive: 16) the same position pair can be reported by several shapes."""
    return sorted(zip(h["query"].tolist(), h["subject"].tolist(), h["seed_offset"].tolist(), h["score"].tolist()))


def blosum62_matrix8():
    """32x32 int8 scoring matrix as the reference holds it (score_matrix.h matrix8), from the swipe tap header."""
    from tapfile import read_tap
    return read_tap(os.path.join(GOLDEN, "swipe_fast.tap"), max_records=1)[0]["matrix8"]


@pytest.mark.parametrize("tap", ["ext_fast.tap", "ext_fast_synth.tap", "ext_6x10.tap", "ext_rank.tap", "ext_default.tap", "ext_default_synth.tap", "ext_sensitive.tap", "ext_blastx.tap"])
def test_seed_stage_hit_multiset_equals_reference(tap):
    cfg, recs = read_ext_tap(os.path.join(GOLDEN, tap))
    assert cfg["ungapped_evalue"] == (10000.0 if "default" in tap or "sensitive" in tap or "blastx" in tap else 0.0) and cfg["index_chunks"] == 4
    c = orc.seed_cfg_from_tap(cfg, blosum62_matrix8())
    hits = orc.seed_search(c, cfg["query"]["data"], cfg["query"]["limits"], cfg["target"]["data"], cfg["target"]["limits"])
    ref = np.concatenate([r["hits"] for r in recs])
    assert len(ref) > 300
    assert len(hits) == len(ref)
    if "sensitive" not in tap:
        assert len(ref) == len(hit_set(ref))                      # few shapes: a multiset without duplicates
    assert hit_multiset(hits) == hit_multiset(ref)
    if "default" in tap:                                        # both batch rules occur: scalar (> 255 kept) and int8 (saturated)
        assert (ref["score"] == 255).any() and (ref["score"] > 255).any() and (ref["score"] < 0xFFFF).all()
    for r in recs:                                                # extend() is called once per query with its own hits only
        assert (r["hits"]["query"] // cfg["query_contexts"] == r["query_id"]).all()


@pytest.mark.parametrize("tap", ["ext_hashed.tap", "ext_hashed_default.tap", "ext_hashed_sens.tap"])
def test_query_indexed_mode_equals_reference(tap):
    """The reference's query-indexed algorithm (--algo 1; what --algo auto picks at BASELINE's C2-C4 sizes) on sequences with
    masked runs (also at the start of sequences), stop codons and ambiguity letters: hashed seed encoding, complexity filter at
    enumeration, one index chunk. Goldens: reference run with --algo 1 (tests/golden/hashed_{q,db}.faa)."""
    cfg, recs = read_ext_tap(os.path.join(GOLDEN, tap))
    assert cfg["index_chunks"] == 1
    c = orc.seed_cfg_from_tap(cfg, blosum62_matrix8())
    c.seed_encoding = 1
    hits = orc.seed_search(c, cfg["query"]["data"], cfg["query"]["limits"], cfg["target"]["data"], cfg["target"]["limits"])
    ref = np.concatenate([r["hits"] for r in recs])
    assert len(ref) > 100
    assert hit_multiset(hits) == hit_multiset(ref)
    c.seed_encoding = 0
    other = orc.seed_search(c, cfg["query"]["data"], cfg["query"]["limits"], cfg["target"]["data"], cfg["target"]["limits"])
    print("spaced-seed rules on the same input differ in %d hits" % len(set(hit_multiset(other)) ^ set(hit_multiset(ref))))


def test_duplicate_query_seeds_with_ambiguity_letters():
    """SURVEY 8 row a5, the corner DESIGN.md used to list as a deviation. B, J and Z reduce to class 0 like A (Reduction's map_ is
    zero-filled, basic.cpp:269), so a window with B at a care position carries the SAME seed as the window with A there -- but
    seed_is_complex answers "not complex" for it (letter >= TRUE_AA, seed_complexity.cpp:43-44). mask_seeds tests the FIRST query
    position of a joined group; the seed stage here tests the SMALLEST. They are the same position: the query seed array of a
    partition is filled in block order (BufferedWriter, seed_array_impl.h:43-93), radix_cluster scatters in input order
    (radix_cluster.h:62-101) and both join variants write a group's values in input order (hash_join.h:98-104,159-165).
    tests/golden/ext_bjz.tap provokes it from the reference itself: 120 query pairs whose 16-letter windows differ only by A vs
    B / J / Z at one care position, in both file orders -- ambiguity letter second: the group is kept and both queries get their
    hit; ambiguity letter first: the whole group is erased (golden minted by make_swipe_golden.sh step 11)."""
    cfg, recs = read_ext_tap(os.path.join(GOLDEN, "ext_bjz.tap"))
    c = orc.seed_cfg_from_tap(cfg, blosum62_matrix8())
    hits = orc.seed_search(c, cfg["query"]["data"], cfg["query"]["limits"], cfg["target"]["data"], cfg["target"]["limits"])
    ref = np.concatenate([r["hits"] for r in recs])
    assert hit_multiset(hits) == hit_multiset(ref) and len(ref) > 100
    # the two arrangements really behave differently in the reference: queries 0..39 (B, complex window first) all have hits,
    # of queries 40..79 (B first) only the few that another seed reaches
    with_hits = {int(r["query_id"]) for r in recs if len(r["hits"])}
    assert all(q in with_hits for q in range(40)) and sum(q in with_hits for q in range(40, 80)) < 12
