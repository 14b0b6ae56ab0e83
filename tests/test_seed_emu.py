"""CPU check of the GPU seed stage's per-thread code (diamond_amd/csrc/seed_core.h) and of its order-free data flow
(query seed table + one reference stream + per-letter mask times), run through tests/emu/seed_emu.cpp, against the
hits tapped from the genuine reference and against the oracle."""
import os
import numpy as np
import pytest

import oracle_py as orc
import emu_py as emu
from tapfile import read_ext_tap
from test_oracle_seed import blosum62_matrix8, hit_multiset

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def hit_set(h):
    return set(zip(h["query"].tolist(), h["subject"].tolist(), h["seed_offset"].tolist(), h["score"].tolist()))


@pytest.mark.parametrize("tap", ["ext_fast.tap", "ext_fast_synth.tap", "ext_6x10.tap", "ext_rank.tap", "ext_default.tap", "ext_default_synth.tap", "ext_sensitive.tap", "ext_blastx.tap"])
def test_emulated_seed_stage_equals_reference_hits(tap):
    cfg, recs = read_ext_tap(os.path.join(GOLDEN, tap))
    c = emu.seed_params_from_tap(cfg)
    hits = emu.seed_search(c, cfg["query"]["data"], cfg["query"]["limits"], cfg["target"]["data"], cfg["target"]["limits"],
                           matrix8=blosum62_matrix8())
    ref = np.concatenate([r["hits"] for r in recs])
    assert len(hits) == len(ref) and hit_multiset(hits) == hit_multiset(ref)


@pytest.mark.parametrize("chunks,bits", [(1, 8), (4, 8), (3, 9), (7, 10)])
def test_emulated_seed_stage_equals_oracle_other_partitionings(chunks, bits):
    """Index-chunk count and seed-partition bits change which seed hit of a diagonal run is the kept (left-most in
    chunk order) one; the order-free GPU formulation must follow the oracle for every setting."""
    cfg, _ = read_ext_tap(os.path.join(GOLDEN, "ext_fast_synth.tap"), max_records=1)
    cfg = dict(cfg, index_chunks=chunks, seedp_bits=bits)
    oc, ec = orc.seed_cfg_from_tap(cfg), emu.seed_params_from_tap(cfg)
    a = orc.seed_search(oc, cfg["query"]["data"], cfg["query"]["limits"], cfg["target"]["data"], cfg["target"]["limits"])
    b = emu.seed_search(ec, cfg["query"]["data"], cfg["query"]["limits"], cfg["target"]["data"], cfg["target"]["limits"])
    assert len(a) == len(b) > 300 and hit_set(a) == hit_set(b)


@pytest.mark.parametrize("tap", ["ext_hashed.tap", "ext_hashed_default.tap", "ext_hashed_sens.tap"])
def test_emulated_query_indexed_mode_equals_reference_hits(tap):
    """The reference's query-indexed algorithm (--algo 1): hashed seed keys (seed_key_hashed), complexity filter and mask when
    the query seeds are indexed, no per-group mask pass -- on sequences with masked runs, stop codons and ambiguity letters."""
    cfg, recs = read_ext_tap(os.path.join(GOLDEN, tap))
    c = emu.seed_params_from_tap(dict(cfg, seed_encoding=1))
    hits = emu.seed_search(c, cfg["query"]["data"], cfg["query"]["limits"], cfg["target"]["data"], cfg["target"]["limits"],
                           matrix8=blosum62_matrix8())
    ref = np.concatenate([r["hits"] for r in recs])
    assert len(hits) == len(ref) and hit_multiset(hits) == hit_multiset(ref)
