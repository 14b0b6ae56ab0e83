"""-m gpu: the device half of the extension stage (csrc/plan_kernels.hip, csrc/extend_kernels.hip; round 6) against the Match lists
the genuine reference's Extension::extend returned (tests/golden/ext_*.tap), with the checks on WHICH path produced the records:
 * the default search of a protein block is planned and extended in HBM (dmnd_extend_plan_stats / dmnd_extend_device_stats);
 * with a trace budget of 8 MB (DMND_TRACE_ARENA_MB, read when the context is made) round 1 sweeps for scores only and round 2
   sweeps the survivors again with traceback -- same records;
 * the same with the row classes of the sweeps forced on (DMND_SWEEP_ROWS=1: small blocks would not take them);
 * a skewed block whose queries have more targets than a ranking chunk (/root/reference/src/align/extend.cpp:79-92: 128 with -k 25)
   is ranked in chunks on the device -- same records as the reference binary on the same files (tests/test_gpu_skew.py holds the
   big case; here a small one that also runs with a tiny budget)."""
import os
import subprocess

import numpy as np
import pytest
import torch

from tapfile import read_ext_tap
from diamond_amd import hip, synth, workload
from test_gpu_seed import to_hip_params

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
REF = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "diamond")
HSP_KEYS = "score q_begin q_end s_begin s_end length identities mismatches gap_openings gaps".split()


def _search(ctx, cfg):
    qd, ql, td, tl = cfg["query"]["data"], cfg["query"]["limits"], cfg["target"]["data"], cfg["target"]["limits"]
    ctx.upload_block(hip.QUERY, qd, ql)
    ctx.upload_block(hip.TARGET, td, tl)
    ctx.set_db_letters(float(tl[-1] - tl[0] - (len(tl) - 1)))
    ctx.set_gapped_filter(cfg["gapped_filter_evalue"])
    ctx.set_query_contexts(cfg["query_contexts"])
    hits = ctx.seed_search(to_hip_params(cfg))
    return ctx.extend(qd, td, hits, threads=4)[0]


def _check(m, recs):
    pos = 0
    for r in sorted(recs, key=lambda x: x["query_id"]):
        for ref in r["matches"]:
            got = m[pos]
            pos += 1
            assert (got["query"], got["target"]) == (r["query_id"], ref["target_block_id"])
            for k in HSP_KEYS:
                assert got["hsp"][k] == ref["hsps"][0][k], (k, r["query_id"])
            assert got["evalue"] == pytest.approx(ref["hsps"][0]["evalue"], rel=1e-6, abs=0)
            assert got["bit_score"] == pytest.approx(ref["hsps"][0]["bit_score"], rel=1e-12)
    assert pos == len(m)


@pytest.mark.parametrize("tap", ["ext_fast_synth.tap", "ext_default_synth.tap", "ext_sensitive.tap", "ext_long.tap"])
@pytest.mark.parametrize("arena_mb,rows", [(None, None), ("8", None), (None, "1"), ("8", "1")])
def test_protein_search_is_extended_in_hbm_and_equals_the_reference(tap, arena_mb, rows, monkeypatch):
    """rows = "1": the row classes of the packed 16-bit sweeps (eight items per wavefront, swipe16_kernels.hip) whatever the number
    of items -- by default only iterations of 131 072 items and more take them (tests/test_gpu_skew.py has such a block)."""
    assert torch.cuda.is_available()
    if arena_mb:
        monkeypatch.setenv("DMND_TRACE_ARENA_MB", arena_mb)
    if rows:
        monkeypatch.setenv("DMND_SWEEP_ROWS", rows)
    ctx = hip.Context()
    try:
        cfg, recs = read_ext_tap(os.path.join(GOLDEN, tap))
        m = _search(ctx, cfg)
        plan, dev = ctx.extend_plan_stats(), ctx.extend_device_stats()
        assert plan["groups"] > 0 and plan["bands"] > 0, "the planner did not run on the device"
        if tap != "ext_long.tap":      # (its 30 000-letter sequences give matrices above max_swipe_dp: those queries take the host path by design)
            assert dev["queries"] > 0 and dev["records"] > 0, "no query was extended on the device"
            # all but the queries handed back (ambiguous e-value order, saturation, groups the planner left to the host) came from HBM
            assert dev["queries_back_to_host"] <= max(1, dev["queries"] // 50)
        _check(m, recs)
    finally:
        ctx.close()


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/diamond missing")
@pytest.mark.parametrize("arena_mb", [None, "8"])
def test_queries_ranked_in_chunks_equal_the_reference_binary(tmp_path, arena_mb):
    """300 queries against 3 families of 400 members: every query has ~400 targets = four ranking chunks of 128. The CLI's records
    (device path; with arena_mb = 8 every chunk is swept for scores only) must equal the reference binary's output on the same files."""
    db, doff, q, qoff = synth.generate(3, members=400, queries=300, seed=11, sub=(0.1, 0.3), qsub=(0.1, 0.3))
    synth.write_fasta(str(tmp_path / "db.faa"), "t", db, doff)
    synth.write_fasta(str(tmp_path / "q.faa"), "q", q, qoff)
    assert subprocess.run([REF, "makedb", "--in", str(tmp_path / "db.faa"), "-d", str(tmp_path / "db")], capture_output=True).returncode == 0
    common = ["blastp", "--fast", "--algo", "0", "--masking", "0", "--motif-masking", "0", "-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.dmnd")]
    r = subprocess.run([REF] + common + ["-o", str(tmp_path / "ref.tsv"), "-p", "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1000:]
    env = dict(os.environ, DMND_TRACE="1")
    if arena_mb:
        env["DMND_TRACE_ARENA_MB"] = arena_mb
    cli = os.path.join(os.path.dirname(HERE), "diamond_amd", "diamond-hip")
    h = subprocess.run([cli] + common + ["-o", str(tmp_path / "hip.tsv")], capture_output=True, text=True, timeout=600, env=env)
    assert h.returncode == 0, h.stderr[-1000:]
    chunks = [l for l in h.stderr.splitlines() if "dmnd_extend (device half): chunk" in l]
    assert any("chunk 1:" in l for l in chunks), "no query was ranked in more than one chunk on the device:\n" + h.stderr[-1500:]
    if arena_mb:
        assert any("scores only" in l for l in chunks)
    assert open(tmp_path / "hip.tsv", "rb").read() == open(tmp_path / "ref.tsv", "rb").read()
    assert os.path.getsize(tmp_path / "ref.tsv") > 10000


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/diamond missing")
@pytest.mark.parametrize("extra", [["-k", "1"], ["-k", "5"], ["-k", "200"], ["--comp-based-stats", "0"], ["-e", "1e-20"], ["--sensitive"]],
                         ids=["k1", "k5", "k200", "cbs0", "e1e-20", "sensitive"])
def test_device_extension_options_equal_the_reference_binary(tmp_path, extra):
    """The options that change what the device half decides -- -k below and above the ranking chunk (with -k 200 the chunk is 224 and
    a first chunk may have to grow: those queries stay on the host), no composition bias, a strict e-value cutoff, the gapped filter of
    --sensitive -- on queries with several ranking chunks and on queries with none: byte-identical to the reference binary."""
    db, doff, q, qoff = synth.generate(40, members=60, queries=300, seed=23, sub=(0.1, 0.4), qsub=(0.1, 0.4))
    synth.write_fasta(str(tmp_path / "db.faa"), "t", db, doff)
    synth.write_fasta(str(tmp_path / "q.faa"), "q", q, qoff)
    assert subprocess.run([REF, "makedb", "--in", str(tmp_path / "db.faa"), "-d", str(tmp_path / "db")], capture_output=True).returncode == 0
    common = ["blastp", "--algo", "0", "--masking", "0", "--motif-masking", "0", "-q", str(tmp_path / "q.faa"), "-d", str(tmp_path / "db.dmnd")] + extra
    if "--sensitive" not in extra:
        common.append("--fast")
    r = subprocess.run([REF] + common + ["-o", str(tmp_path / "ref.tsv"), "-p", "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1000:]
    cli = os.path.join(os.path.dirname(HERE), "diamond_amd", "diamond-hip")
    h = subprocess.run([cli] + common + ["-o", str(tmp_path / "hip.tsv")], capture_output=True, text=True, timeout=600, env=dict(os.environ, DMND_TRACE="1"))
    assert h.returncode == 0, h.stderr[-1000:]
    assert "dmnd_extend (device half)" in h.stderr
    assert open(tmp_path / "hip.tsv", "rb").read() == open(tmp_path / "ref.tsv", "rb").read()
    assert os.path.getsize(tmp_path / "ref.tsv") > 1000
