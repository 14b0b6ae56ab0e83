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


@pytest.mark.parametrize("tap", ["ext_fast.tap", "ext_fast_synth.tap", "ext_6x10.tap", "ext_rank.tap"])
def test_seed_stage_hit_multiset_equals_reference(tap):
    cfg, recs = read_ext_tap(os.path.join(GOLDEN, tap))
    assert cfg["ungapped_evalue"] == 0.0 and cfg["index_chunks"] == 4
    c = orc.seed_cfg_from_tap(cfg)
    hits = orc.seed_search(c, cfg["query"]["data"], cfg["query"]["limits"], cfg["target"]["data"], cfg["target"]["limits"])
    ref = np.concatenate([r["hits"] for r in recs])
    assert len(ref) > 300
    assert len(hits) == len(ref) == len(hit_set(hits))          # a multiset without duplicates
    assert hit_set(hits) == hit_set(ref)
    for r in recs:                                                # extend() is called once per query with its own hits only
        assert (r["hits"]["query"] == r["query_id"]).all()
