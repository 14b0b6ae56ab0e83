"""-m gpu: dmnd_xdrop_ungapped (the optional entry): the x-drop ungapped extension of every seed hit of a block pair in
one launch (xdrop_seg_kernel, the extension stage's own first device step) against a plain restatement of the reference function
(/root/reference/src/dp/ungapped_align.cpp:151-199: the left walk from qa - 1 / sa - 1, the right walk from qa / sa, both ending at a
delimiter or xdrop below the best; DiagonalSegment(qa - delta, sa - delta, len + delta, score)), on the seed hits the reference itself
handed to Extension::extend (tests/golden/ext_fast_synth.tap, ext_default.tap), without and with the Hauser composition bias (the bias
values come from the host restatement behind dmnd_extend_plan, which the reference's taps pin)."""
import os
import numpy as np
import pytest

from tapfile import read_ext_tap
from diamond_amd import hip

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _reference_xdrop(M, q, cbs, t, qa, sa, xdrop):
    score = st = 0
    n, delta, length = 1, 0, 0
    i, j = qa - 1, sa - 1
    while score - st < xdrop and (q[i] & 31) != 31 and (t[j] & 31) != 31:
        st += int(M[q[i] & 31, t[j] & 31]) + (int(cbs[i]) if cbs is not None else 0)
        if st > score:
            score, delta = st, n
        i -= 1; j -= 1; n += 1
    i, j, st, n = qa, sa, score, 1
    while score - st < xdrop and (q[i] & 31) != 31 and (t[j] & 31) != 31:
        st += int(M[q[i] & 31, t[j] & 31]) + (int(cbs[i]) if cbs is not None else 0)
        if st > score:
            score, length = st, n
        i += 1; j += 1; n += 1
    return delta, length + delta, score


@pytest.mark.parametrize("tap", ["ext_fast_synth.tap", "ext_default.tap"])
def test_xdrop_ungapped_equals_the_reference_function(tap):
    cfg, recs = read_ext_tap(os.path.join(GOLDEN, tap))
    qd, ql, td, tl = cfg["query"]["data"], cfg["query"]["limits"], cfg["target"]["data"], cfg["target"]["limits"]
    src = np.concatenate([r["hits"] for r in recs])
    rng = np.random.default_rng(1)
    src = src[rng.permutation(len(src))[:600]]
    hits = np.zeros(len(src), dtype=hip.SEED_HIT_DTYPE)
    for f in ("query", "seed_offset", "subject", "score"):
        hits[f] = src[f]
    params = hip.default_params()
    M = hip.matrix_of(params)
    cbs, _ = hip.extend_plan(params, qd, ql, td, tl, np.zeros(0, hip.SEED_HIT_DTYPE))
    ctx = hip.Context(params=params)
    try:
        ctx.upload_block(hip.QUERY, qd, ql)
        ctx.upload_block(hip.TARGET, td, tl)
        plain = ctx.xdrop_ungapped(hits)
        biased = ctx.xdrop_ungapped(hits, use_bias=True)
        narrow = ctx.xdrop_ungapped(hits, xdrop=7)
        assert len(ctx.xdrop_ungapped(hits[:0])) == 0
    finally:
        ctx.close()
    differ = 0
    for k, h in enumerate(hits):
        qa, sa = int(ql[h["query"]]) + int(h["seed_offset"]), int(h["subject"])
        for got, bias, x in ((plain[k], None, 20), (biased[k], cbs, 20), (narrow[k], None, 7)):
            delta, length, score = _reference_xdrop(M, qd, bias, td, qa, sa, x)
            assert (int(got["i"]), int(got["j"]), int(got["len"]), int(got["score"])) == (int(h["seed_offset"]) - delta, sa - delta, length, score), (k, x, bias is not None)
        differ += int(plain[k]["score"] != biased[k]["score"]) + int(plain[k]["len"] != narrow[k]["len"])
    assert differ > 20                                           # the bias and the x-drop value matter on these hits
