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


def test_filter_off_and_argument_errors(ctx):
    ctx.set_gapped_filter(0.0)
    with pytest.raises(hip.DiamondHipError):
        ctx.gapped_filter(np.zeros(1, dtype=hip.SEED_HIT_DTYPE))
