"""-m gpu end-to-end parity of the MI355X hot path through the C ABI: seed stage (dmnd_seed_search) -> extension stage
(dmnd_extend: host chaining + two batched GPU Smith-Waterman rounds + culling) -> BLAST tabular text, against
 (a) the Match lists the genuine reference's Extension::extend returned for the same queries (tests/golden/ext_*.tap),
 (b) the reference's own TSV output, byte for byte (tests/golden/fast_synth.tsv)."""
import os
import numpy as np
import pytest
import torch

from tapfile import read_ext_tap
from diamond_amd import hip
from test_gpu_seed import to_hip_params

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HSP_KEYS = "score q_begin q_end s_begin s_end length identities mismatches gap_openings gaps".split()


@pytest.fixture(scope="module")
def ctx():
    assert torch.cuda.is_available()
    c = hip.Context()
    yield c
    c.close()


def _run(ctx, cfg, db_letters):
    qd, ql, td, tl = cfg["query"]["data"], cfg["query"]["limits"], cfg["target"]["data"], cfg["target"]["limits"]
    ctx.upload_block(hip.QUERY, qd, ql)
    ctx.upload_block(hip.TARGET, td, tl)
    ctx.set_db_letters(db_letters)
    ctx.set_gapped_filter(cfg["gapped_filter_evalue"])          # 0 except --sensitive (1.0)
    ctx.set_query_contexts(cfg["query_contexts"])               # 6 for the blastx golden
    hits = ctx.seed_search(to_hip_params(cfg))
    return ctx.extend(qd, td, hits, threads=4)[0]


@pytest.mark.parametrize("tap", ["ext_fast_synth.tap", "ext_fast.tap", "ext_6x10.tap", "ext_rank.tap", "ext_default.tap", "ext_default_synth.tap", "ext_sensitive.tap", "ext_blastx.tap", "ext_long.tap"])
def test_matches_equal_reference_extend(ctx, tap):
    cfg, recs = read_ext_tap(os.path.join(GOLDEN, tap))
    tl = cfg["target"]["limits"]
    db_letters = float(tl[-1] - tl[0] - (len(tl) - 1))
    m = _run(ctx, cfg, db_letters)
    pos = 0
    n = 0
    for r in sorted(recs, key=lambda x: x["query_id"]):          # tap records are in call order of the reference's worker threads
        for ref in r["matches"]:
            assert pos < len(m)
            got = m[pos]
            pos += 1
            assert (got["query"], got["target"]) == (r["query_id"], ref["target_block_id"])
            h = ref["hsps"][0]
            assert len(ref["hsps"]) == 1
            for k in HSP_KEYS:
                assert got["hsp"][k] == h[k], (k, r["query_id"], ref["target_block_id"])
            assert got["frame"] == h["frame"]
            assert got["evalue"] == pytest.approx(h["evalue"], rel=1e-6, abs=0)      # north_star tolerance
            assert got["bit_score"] == pytest.approx(h["bit_score"], rel=1e-12)
            assert got["ungapped_score"] == ref["ungapped_score"]
            n += 1
    assert pos == len(m) and n > (300 if tap != "ext_long.tap" else 4)      # ext_long: 9 kb proteins, DP > 1e6 cells -> statistics-without-traceback path


@pytest.mark.parametrize("tap,tsv", [("ext_fast_synth.tap", "fast_synth.tsv"), ("ext_rank.tap", "rank.tsv"),
                                     ("ext_default_synth.tap", "default_synth.tsv"), ("ext_default.tap", "default.tsv"),
                                     ("ext_sensitive.tap", "sensitive.tsv"), ("ext_blastx.tap", "blastx.tsv"), ("ext_long.tap", "long.tsv")])
def test_tabular_output_is_byte_identical_to_reference(ctx, tap, tsv):
    cfg, recs = read_ext_tap(os.path.join(GOLDEN, tap))
    tl = cfg["target"]["limits"]
    m = _run(ctx, cfg, float(tl[-1] - tl[0] - (len(tl) - 1)))
    if tsv in ("default.tsv", "sensitive.tsv"):                                       # the reference's own fixture (src/test/data.faa) against itself
        qids = tids = open(os.path.join(GOLDEN, "data_ids.txt")).read().split()
    else:
        qids = ["q%d" % i for i in range(cfg["query"]["n"])]
        tids = ["t%d" % i for i in range(cfg["target"]["n"])]
    source_lens = None
    if tsv == "blastx.tsv":                                        # DNA coordinates need the read lengths
        reads = [l.strip() for l in open(os.path.join(GOLDEN, "blastx_reads.fna")) if not l.startswith(">")]
        source_lens = [len(x) for x in reads]
        qids = ["r%d" % i for i in range(len(reads))]
        assert len(reads) * 6 == cfg["query"]["n"]
    text = hip.format_tab(m, qids, tids, source_lens)
    ref = open(os.path.join(GOLDEN, tsv)).read()
    assert len(ref.splitlines()) > (300 if tsv != "long.tsv" else 4)
    assert text == ref


def test_transcripts_equal_reference(ctx):
    """dmnd_extend with a transcript arena (the unsplit path): the packed edit transcript of every reported alignment equals
    the reference's (PackedOperation bytes, basic/packed_transcript.h)."""
    cfg, recs = read_ext_tap(os.path.join(GOLDEN, "ext_fast_synth.tap"))
    qd, ql, td, tl = cfg["query"]["data"], cfg["query"]["limits"], cfg["target"]["data"], cfg["target"]["limits"]
    ctx.upload_block(hip.QUERY, qd, ql)
    ctx.upload_block(hip.TARGET, td, tl)
    ctx.set_db_letters(float(tl[-1] - tl[0] - (len(tl) - 1)))
    ctx.set_gapped_filter(0.0)
    ctx.set_query_contexts(1)
    hits = ctx.seed_search(to_hip_params(cfg))
    m, tr = ctx.extend(qd, td, hits, threads=4, with_transcripts=True)
    m0 = ctx.extend(qd, td, hits, threads=4)[0]
    for k in HSP_KEYS:
        assert np.array_equal(m["hsp"][k], m0["hsp"][k])                         # same alignments with and without transcripts
    pos = 0
    for r in sorted(recs, key=lambda x: x["query_id"]):
        for ref in r["matches"]:
            got = m[pos]
            pos += 1
            h = ref["hsps"][0]
            want = np.asarray(h["transcript"], np.uint8)
            want = want[:-1] if len(want) and want[-1] == 0 else want
            off, n = int(got["hsp"]["transcript_off"]), int(got["hsp"]["transcript_len"])
            assert off >= 0 and n == len(want) and np.array_equal(tr[off:off + n], want)
            assert tr[off + n] == 0
    assert pos == len(m) > 300
