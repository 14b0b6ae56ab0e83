"""CPU test of the host half of the extension stage (diamond_amd/csrc/extend_host.hip + chain_host.h, through the C ABI
entry dmnd_extend_plan): Hauser composition bias, load_hits, x-drop ungapped extension, greedy chaining, band merging.
Known answers: the DpTargets (band geometry) and bias vectors the genuine reference passed to DP::BandedSwipe::swipe
for the same queries and seed hits (both taps in one run: tests/golden/ext_fast_synth.tap + swipe_fast_synth.tap)."""
import os
import numpy as np

from tapfile import read_tap, read_ext_tap
from diamond_amd import hip

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load():
    cfg, ext = read_ext_tap(os.path.join(GOLDEN, "ext_fast_synth.tap"))
    hdr, sw = read_tap(os.path.join(GOLDEN, "swipe_fast_synth.tap"))
    return cfg, ext, hdr, sw


def test_band_geometry_and_hauser_equal_reference():
    cfg, ext, hdr, sw = _load()
    qd, ql, td, tl = cfg["query"]["data"], cfg["query"]["limits"], cfg["target"]["data"], cfg["target"]["limits"]
    hits = np.concatenate([r["hits"] for r in ext])
    h = np.zeros(hits.size, dtype=hip.SEED_HIT_DTYPE)
    for k in ("query", "seed_offset", "subject", "score"):
        h[k] = hits[k]
    h = h[np.argsort(h["query"], kind="stable")]
    p = hip.default_params()
    p.db_letters = hdr["db_letters"]
    cbs, plan = hip.extend_plan(p, qd, ql, td, tl, h, threads=2)
    # reference side: round-1 calls (hsp_values == 0), keyed by the query's letters
    by_query = {}
    for i in range(len(ql) - 1):
        by_query[qd[ql[i]: ql[i + 1] - 1].tobytes()] = i
    seen = set()
    n_targets = 0
    for rec in sw:
        if rec["hsp_values"] != 0:
            continue
        qi = by_query[rec["query"].tobytes()]
        assert qi not in seen          # one round-1 call per query (single ranking chunk)
        seen.add(qi)
        assert np.array_equal(rec["cbs"], cbs[ql[qi]: ql[qi] + len(rec["query"])]), qi
        want = sorted((t["seq"].tobytes(), t["d_begin"], t["d_end"]) for t in rec["targets"])
        mine = plan[plan["query"] == qi]
        got = sorted((td[tl[t]: tl[t + 1] - 1].tobytes(), int(a), int(b)) for t, a, b in zip(mine["target"], mine["d_begin"], mine["d_end"]))
        assert got == want, qi
        n_targets += len(want)
    assert n_targets > 300
    # queries that produced DpTargets in our plan but no swipe call in the reference would be a bug, too
    assert set(np.unique(plan["query"]).tolist()) == seen
