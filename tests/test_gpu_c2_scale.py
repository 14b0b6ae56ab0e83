"""-m gpu: the whole hot path at BASELINE config C2 size (blastp --fast, 10k queries x 1M-sequence / 3.0e8-letter DB)
through the C ABI, checked with size-independent properties (no per-item oracle is feasible at this size):
determinism, ordering and top-k bounds of the reported records, ground truth of the synthetic families, and a sampled
re-derivation of reported alignments by the CPU oracle on the same band and composition bias."""
import numpy as np
import pytest
import torch

import oracle_py as orc
from diamond_amd import hip, synth, workload

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]
KEYS = "score q_begin q_end s_begin s_end length identities mismatches gap_openings gaps".split()


def test_c2_full_path_properties():
    assert torch.cuda.is_available()
    members = 10
    db, doff, q, qoff, fam = synth.generate(100_000, members=members, queries=10_000, seed=20260923, family=True)
    qd, ql = workload.sequence_set(q, qoff)
    td, tl = workload.sequence_set(db, doff)
    params = hip.default_params()
    params.db_letters = float(doff[-1])
    ctx = hip.Context(params=params)
    try:
        ctx.upload_block(hip.QUERY, qd, ql)
        ctx.upload_block(hip.TARGET, td, tl)
        sp = hip.seed_params_fast(threads=8)
        hits = ctx.seed_search(sp)
        m, _ = ctx.extend(qd, td, hits, threads=16)
        # determinism of the whole path
        hits2 = ctx.seed_search(sp)
        m2, _ = ctx.extend(qd, td, hits2, threads=7)                  # a different host thread count must not matter
        assert np.array_equal(hits, hits2) and np.array_equal(m, m2)
    finally:
        ctx.close()
    assert len(hits) > 15_000 and len(m) > 8_000
    # every hit is a seed match between its two sequences (positions inside the sequences, same reduced letters)
    pos = np.array([sp.shape_pos[0][k] for k in range(sp.shape_weight[0])])
    red = np.array([sp.reduction[i] for i in range(32)])
    qloc = ql[hits["query"]] + hits["seed_offset"]
    assert (red[qd[qloc[:, None] + pos[None, :]] & 31] == red[td[hits["subject"][:, None] + pos[None, :]] & 31]).all()
    # records: grouped by query ascending, <= 25 per query, ordered by (evalue asc, score desc, target asc), below the e-value cutoff
    qv = m["query"].astype(np.int64)
    assert (np.diff(qv) >= 0).all()
    _, counts = np.unique(qv, return_counts=True)
    assert counts.max() <= 25
    same = qv[1:] == qv[:-1]
    ev, sc, tg = m["evalue"], m["hsp"]["score"], m["target"].astype(np.int64)
    ordered = (ev[1:] > ev[:-1]) | ((ev[1:] == ev[:-1]) & ((sc[1:] < sc[:-1]) | ((sc[1:] == sc[:-1]) & (tg[1:] > tg[:-1]))))
    assert ordered[same].all()
    assert (ev <= 0.001).all() and (sc > 0).all()
    # every reported target received at least one seed hit of that query
    tid = np.searchsorted(tl, hits["subject"], side="right") - 1
    pairs = set(zip(hits["query"].tolist(), tid.tolist()))
    assert all((int(a), int(b)) in pairs for a, b in zip(m["query"], m["target"]))
    # ground truth of the generator: the best hit of a query derived from family f is a member of f (decoys align to nothing much)
    first = np.r_[True, qv[1:] != qv[:-1]]
    top_q, top_t = qv[first], tg[first]
    derived = fam[top_q] >= 0
    assert derived.sum() > 4000
    assert ((top_t[derived] // members) == fam[top_q[derived]]).mean() > 0.99
    # sampled parity: oracle on the same band + bias reproduces score, coordinates and statistics; e-value within 1e-6
    rng = np.random.default_rng(3)
    sample = rng.choice(len(m), 150, replace=False)
    e = orc.evaluer(float(doff[-1]))
    M = hip.matrix_of(params)
    cbs_cache = {}
    for k in sample:
        r = m[k]
        qi, ti = int(r["query"]), int(r["target"])
        qs, ts = qd[ql[qi]:ql[qi + 1] - 1], td[tl[ti]:tl[ti + 1] - 1]
        if qi not in cbs_cache:
            one_ql = np.array([256, 256 + len(qs) + 1], np.int64)
            one_qd = np.concatenate([np.full(256, 31, np.int8), qs, np.full(257, 31, np.int8)])
            cbs_cache[qi] = hip.extend_plan(params, one_qd, one_ql, one_qd, one_ql, np.zeros(0, hip.SEED_HIT_DTYPE))[0][256:256 + len(qs)]
        rc, o, _ = orc.banded_swipe(qs, cbs_cache[qi], ts, int(r["d_begin"]), int(r["d_end"]), M, 11, 1, orc.TRACEBACK)
        assert rc == 0
        for key in KEYS:
            assert o[key] == r["hsp"][key], (key, qi, ti)
        assert r["evalue"] == pytest.approx(orc.evalue(e, o["score"], len(qs), len(ts)), rel=1e-6, abs=0)


def _family_truth(m, fam, members, min_frac):
    qv, tg = m["query"].astype(np.int64), m["target"].astype(np.int64)
    first = np.r_[True, qv[1:] != qv[:-1]]
    top_q, top_t = qv[first], tg[first]
    derived = fam[top_q] >= 0
    assert derived.sum() > 1000
    assert ((top_t[derived] // members) == fam[top_q[derived]]).mean() > min_frac
    _, counts = np.unique(qv, return_counts=True)
    assert counts.max() <= 25 and (m["evalue"] <= 0.001).all()


def test_c3_sensitive_and_c4_blastx_at_scale(capsys):
    """BASELINE configs C3 (blastp --sensitive, 10k x 1M) and C4 (blastx, 5k reads of ~1 kb x 1M): the same database block,
    ground truth of the synthetic families, bounds on the records, and that more sensitive modes find a superset of aligned
    queries. Timings are printed for DESIGN.md (not a bench line)."""
    import time
    assert torch.cuda.is_available()
    members = 10
    db, doff, q, qoff, fam = synth.generate(100_000, members=members, queries=10_000, seed=20260923, family=True)
    qd, ql = workload.sequence_set(q, qoff)
    td, tl = workload.sequence_set(db, doff)
    params = hip.default_params()
    params.db_letters = float(doff[-1])
    ctx = hip.Context(params=params)
    out = {}
    try:
        ctx.upload_block(hip.TARGET, td, tl)
        ctx.upload_block(hip.QUERY, qd, ql)
        aligned = {}
        for name in ("fast", "default", "sensitive"):
            sp, gf = hip.seed_params_preset(name, params, threads=8)
            ctx.set_gapped_filter(gf)
            t0 = time.perf_counter()
            hits = ctx.seed_search(sp)
            t1 = time.perf_counter()
            m, _ = ctx.extend(qd, td, hits, threads=16)
            t2 = time.perf_counter()
            _family_truth(m, fam, members, 0.99)
            aligned[name] = set(m["query"].tolist())
            out[name] = dict(hits=int(len(hits)), matches=int(len(m)), aligned=len(aligned[name]), seed_ms=(t1 - t0) * 1e3,
                             seed_kernel_ms=ctx.seed_kernel_ms(), extend_ms=(t2 - t1) * 1e3, gapped_filter_ms=ctx.gapped_filter_ms(), ext=ctx.extend_stats())
        assert len(aligned["fast"] - aligned["default"]) <= 0.02 * len(aligned["fast"])
        assert len(aligned["default"] - aligned["sensitive"]) <= 0.02 * len(aligned["default"])
        assert len(aligned["sensitive"]) > len(aligned["fast"])
        # C4: 5k reads back-translated from the first 5k queries, six frames each
        n_reads = 5000
        dna, off = synth.back_translate(q[:qoff[n_reads]], qoff[:n_reads + 1], seed=5)
        t0 = time.perf_counter()
        xd, xl = hip.translated_block(dna, off)
        t1 = time.perf_counter()
        ctx.upload_block(hip.QUERY, xd, xl)
        ctx.set_query_contexts(6)
        sp, gf = hip.seed_params_preset("default", params, threads=8)
        sp.query_translated = 1
        ctx.set_gapped_filter(gf)
        hits = ctx.seed_search(sp)
        t2 = time.perf_counter()
        m, _ = ctx.extend(xd, td, hits, threads=16)
        t3 = time.perf_counter()
        _family_truth(m, fam[:n_reads], members, 0.99)
        assert set(np.unique(m["frame"]).tolist()) <= set(range(6)) and len(np.unique(m["frame"])) == 6      # both strands, all offsets
        out["blastx"] = dict(reads=n_reads, hits=int(len(hits)), matches=int(len(m)), aligned=len(set(m["query"].tolist())),
                             translate_host_ms=(t1 - t0) * 1e3, seed_ms=(t2 - t1) * 1e3, seed_kernel_ms=ctx.seed_kernel_ms(), extend_ms=(t3 - t2) * 1e3)
        # the same reads as proteins align to the same top target in (almost) every case
        assert len(set(m["query"].tolist()) ^ set(x for x in aligned["default"] if x < n_reads)) < 0.05 * n_reads
    finally:
        ctx.close()
    with capsys.disabled():
        import json
        print("\\nSCALE_TIMINGS " + json.dumps(out))


def test_c2_tantan_masking_at_scale(capsys):
    """tantan over the 3.0e8-letter C2 reference block in one call: sampled sequences equal the oracle bit for bit, untouched
    delimiters, and the kernel time is printed for DESIGN.md."""
    import json
    from test_oracle_seed import blosum62_matrix8
    assert torch.cuda.is_available()
    db, doff, q, qoff = synth.generate(100_000, members=10, queries=10_000, seed=20260923)
    rng = np.random.default_rng(5)
    db = db.copy()
    for i in rng.choice(len(doff) - 1, 20_000, replace=False):                    # plant repeats in 2 % of the sequences
        n = int(doff[i + 1] - doff[i])
        if n < 90:
            continue
        a, L = int(rng.integers(0, n - 70)), int(rng.integers(25, 70))
        db[doff[i] + a: doff[i] + a + L] = np.resize(rng.integers(0, 20, int(rng.integers(1, 6))).astype(np.int8), L)
    td, tl = workload.sequence_set(db, doff)
    assert hip.load().dmnd_init(0) == 0            # as the command line does: code objects on the device, every kernel launched once
    ctx = hip.Context()
    try:
        ctx.upload_block(hip.TARGET, td, tl)
        masked = td.copy()
        n = ctx.mask_block(hip.TARGET, masked)
        ms = ctx.mask_kernel_ms()
    finally:
        ctx.close()
    assert n > 300_000
    assert (masked[tl[1:] - 1] == 31).all() and (masked[:256] == 31).all()
    changed = masked != td
    assert (masked[changed] == 23).all()
    lr = orc.tantan_matrix(blosum62_matrix8())
    planted = np.nonzero(np.add.reduceat(changed.astype(np.int64), tl[:-1]) > 0)[0]
    sample = np.concatenate([rng.choice(planted, 150, replace=False), rng.choice(len(tl) - 1, 150, replace=False)])
    for i in sample:
        a, b = int(tl[i]), int(tl[i + 1]) - 1
        assert np.array_equal(masked[a:b], orc.tantan_mask(td[a:b], lr)[0]), int(i)
    with capsys.disabled():
        print("\nMASK_TIMING " + json.dumps(dict(letters=int(doff[-1]), masked=int(n), kernel_ms=ms)))
