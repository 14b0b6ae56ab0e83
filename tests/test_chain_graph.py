"""CPU check of the product's host-side chaining (diamond_amd/csrc/chain_graph.h: x-drop extension of seed hits, segment graph,
chain walk, chain joining) against the retired statement-level restatement of the reference (oracle/chain_ref.h), which
reproduced the reference's DpTargets on all goldens: same segments and same chains (diagonal range, score, query and subject
ranges, in the same order) on thousands of random seed-hit sets over related, repetitive and unrelated sequence pairs.
(tests/test_extend_plan.py pins the end result -- the bands -- on the reference's own DpTargets.)"""
import ctypes
import os
import numpy as np

import emu_py as emu
from tapfile import read_tap

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _run(which, q, cbs, t, M, hits, go=11, ge=1):
    pad = 64
    qb = np.full(len(q) + 2 * pad, 31, np.int8); qb[pad:pad + len(q)] = q
    tb = np.full(len(t) + 2 * pad, 31, np.int8); tb[pad:pad + len(t)] = t
    cb = None
    if cbs is not None:
        cb = np.zeros(len(q) + 2 * pad, np.int8); cb[pad:pad + len(q)] = cbs
    hi = np.ascontiguousarray([h[0] for h in hits], dtype=np.int32)
    hj = np.ascontiguousarray([h[1] for h in hits], dtype=np.int32)
    segs = np.zeros(4 * 8192, np.int32)
    chains = np.zeros(7 * 1024, np.int32)
    nseg = ctypes.c_int(0)
    m = np.ascontiguousarray(M, dtype=np.int8)
    n = emu.lib().emu_chain(which, ctypes.c_void_p(qb.ctypes.data + pad), len(q), ctypes.c_void_p(cb.ctypes.data + pad) if cb is not None else None,
                            ctypes.c_void_p(tb.ctypes.data + pad), len(t), m.ctypes.data_as(ctypes.c_void_p), go, ge,
                            hi.ctypes.data_as(ctypes.c_void_p), hj.ctypes.data_as(ctypes.c_void_p), len(hits),
                            segs.ctypes.data_as(ctypes.c_void_p), 8192, ctypes.byref(nseg), chains.ctypes.data_as(ctypes.c_void_p), 1024)
    if n < 0:                                   # which == 2 only: the target does not fit the fixed-capacity instance
        return segs[:4 * nseg.value].reshape(-1, 4).copy(), None
    return segs[:4 * nseg.value].reshape(-1, 4).copy(), chains[:7 * n].reshape(-1, 7).copy()


def _pair(rng, kind):
    qlen = int(rng.integers(30, 400))
    q = rng.integers(0, 20, qlen).astype(np.int8)
    if kind == 0:                               # homolog with substitutions and indels
        t = q.copy()
        mut = rng.random(qlen) < rng.uniform(0.1, 0.5)
        t[mut] = rng.integers(0, 20, int(mut.sum()))
        for _ in range(int(rng.integers(0, 5))):
            cut = int(rng.integers(0, len(t)))
            if rng.random() < 0.5:
                t = np.concatenate([t[:cut], rng.integers(0, 20, int(rng.integers(1, 12))).astype(np.int8), t[cut:]])
            else:
                t = np.concatenate([t[:cut], t[cut + int(rng.integers(1, 12)):]])
        if len(t) < 10:
            t = q.copy()
    elif kind == 1:                             # tandem repeats: many overlapping segments on many diagonals
        unit = rng.integers(0, 20, int(rng.integers(3, 25))).astype(np.int8)
        q = np.resize(unit, qlen).copy()
        t = np.resize(unit, int(rng.integers(30, 400))).copy()
        for s in (q, t):
            mut = rng.random(len(s)) < 0.08
            s[mut] = rng.integers(0, 20, int(mut.sum()))
    elif kind == 2:                             # two domains in swapped order + an unrelated stretch
        a, b = q[:qlen // 2], q[qlen // 2:]
        t = np.concatenate([b, rng.integers(0, 20, int(rng.integers(0, 60))).astype(np.int8), a])
    else:
        t = rng.integers(0, 20, int(rng.integers(30, 400))).astype(np.int8)
    return q, t.astype(np.int8)


def _hits(rng, q, t, w=5):
    # seed hits = exact w-mer matches (subsampled), sorted by (diagonal, j) as the extension stage sorts them
    idx = {}
    for j in range(len(t) - w + 1):
        idx.setdefault(bytes(t[j:j + w]), []).append(j)
    hits = []
    for i in range(len(q) - w + 1):
        for j in idx.get(bytes(q[i:i + w]), ()):
            if rng.random() < 0.7:
                hits.append((i, j))
    if len(hits) < 2:
        hits += [(int(rng.integers(0, len(q))), int(rng.integers(0, len(t)))) for _ in range(3)]
    hits.sort(key=lambda h: (h[0] - h[1], h[1]))
    return hits[:4000]


def test_chains_equal_the_reference_restatement():
    hdr, _ = read_tap(os.path.join(GOLDEN, "swipe_fast.tap"), max_records=1)
    M = hdr["matrix8"]
    rng = np.random.default_rng(2026)
    n_multi = n_chains = 0
    for it in range(1500):
        q, t = _pair(rng, it % 4)
        cbs = rng.integers(-2, 2, len(q)).astype(np.int8) if it % 3 == 0 else None
        hits = _hits(rng, q, t, w=(3, 4, 5)[it % 3])
        s0, c0 = _run(0, q, cbs, t, M, hits)
        s1, c1 = _run(1, q, cbs, t, M, hits)
        assert np.array_equal(s0, s1), it
        assert np.array_equal(c0, c1), (it, c0, c1)
        n_multi += len(s0) > 1
        n_chains += len(c0)
    assert n_multi > 800 and n_chains > 1500


def test_many_segments_take_the_length_cap_path():
    """More than 200 segments: the score-sorted cut by total length (chaining_len_cap / chaining_min_nodes)."""
    hdr, _ = read_tap(os.path.join(GOLDEN, "swipe_fast.tap"), max_records=1)
    M = hdr["matrix8"]
    rng = np.random.default_rng(7)
    seen = 0
    for it in range(12):
        unit = rng.integers(0, 20, int(rng.integers(40, 200))).astype(np.int8)
        q = np.resize(unit, 700).copy()
        t = np.resize(unit, 900).copy()
        for s in (q, t):
            mut = rng.random(len(s)) < 0.33
            s[mut] = rng.integers(0, 20, int(mut.sum()))
        hits = _hits(rng, q, t, w=2)
        s0, c0 = _run(0, q, None, t, M, hits)
        s1, c1 = _run(1, q, None, t, M, hits)
        assert np.array_equal(s0, s1) and np.array_equal(c0, c1), it
        seen += len(s0) > 200
    assert seen >= 3


def test_fixed_capacity_instance_equals_the_host_instance():
    """The device planner's instance of the same source (ChainWorkspaceT<FixedChainPolicy>: arrays of 16 segments / 96 links in a
    lane's private memory, insertion sort) against the host's std::vector instance: identical chains whenever the target fits,
    and most multi-segment targets do fit (the others are chained on the host)."""
    hdr, _ = read_tap(os.path.join(GOLDEN, "swipe_fast.tap"), max_records=1)
    M = hdr["matrix8"]
    rng = np.random.default_rng(606)
    fit = multi = 0
    for it in range(3000):
        q, t = _pair(rng, it % 4)
        cbs = rng.integers(-2, 2, len(q)).astype(np.int8) if it % 3 == 0 else None
        hits = _hits(rng, q, t, w=(4, 5, 6)[it % 3])
        if it % 2:                              # fewer hits: the sizes the seed stage really produces per target
            hits = hits[::max(1, len(hits) // int(rng.integers(2, 14)))]
        s0, c0 = _run(0, q, cbs, t, M, hits)
        s2, c2 = _run(2, q, cbs, t, M, hits)
        assert np.array_equal(s0, s2), it
        if len(s0) > 1:
            multi += 1
        if c2 is None:
            assert len(s0) > 1
            continue
        fit += len(s0) > 1
        assert np.array_equal(c0, c2), (it, c0, c2)
    assert multi > 1500 and fit > 0.5 * multi, (fit, multi)
