"""The exchange arithmetic of dmnd_join_ranks (csrc/rank_join.hip: owner_of, send / receive offsets, the owner-ordered copy) without
a device: dmnd_join_ranks_plan runs the very code the RCCL path runs, the test moves bytes by its plan for 2 ... 8 ranks with uneven
(and empty) shares and checks that every record lands, once, in the query range of its owner, in the order the merge relies on.
The merge it replaces: /root/reference/src/output/join_blocks.cpp:129-199 (records of all blocks, grouped by query)."""
import ctypes

import numpy as np
import pytest

from diamond_amd import hip, multigpu


def plan(lib, queries, n_queries):
    n = len(queries)
    counts = (ctypes.c_int64 * n)(*[len(q) for q in queries])
    arrs = [np.ascontiguousarray(q, dtype=np.uint32) for q in queries]
    qptrs = (ctypes.c_void_p * n)(*[a.ctypes.data if a.size else None for a in arrs])
    cnt = np.zeros((n, n), dtype=np.int64)
    send = np.zeros((n, n), dtype=np.int64)
    recv = np.zeros((n, n), dtype=np.int64)
    n_recv = np.zeros(n, dtype=np.int64)
    place = [np.full(len(q), -1, dtype=np.int64) for q in queries]
    pptrs = (ctypes.c_void_p * n)(*[p.ctypes.data if p.size else None for p in place])
    lib.dmnd_join_ranks_plan.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    rc = lib.dmnd_join_ranks_plan(n, counts, qptrs, n_queries, cnt.ctypes.data, send.ctypes.data, recv.ctypes.data, n_recv.ctypes.data, pptrs)
    return rc, cnt, send, recv, n_recv, place


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 7, 8])
@pytest.mark.parametrize("n_queries", [1, 5, 997, 10000])
def test_every_record_lands_once_in_its_owners_range(n, n_queries):
    lib = hip.load()
    rng = np.random.default_rng(100 * n + n_queries)
    # uneven shares: some sources empty, one holding most of the records, queries drawn with a skew
    sizes = [0 if g % 3 == 1 and n > 2 else int(rng.integers(0, 400)) for g in range(n)]
    sizes[rng.integers(0, n)] += 3000
    queries = [np.sort(np.minimum((rng.random(s) ** 2 * n_queries).astype(np.int64), n_queries - 1)).astype(np.uint32) for s in sizes]
    rc, cnt, send, recv, n_recv, place = plan(lib, queries, n_queries)
    assert rc == 0
    ranges = [multigpu.shard_range(n_queries, n, g) for g in range(n)]
    owner = lambda q: next(g for g, (b, e) in enumerate(ranges) if b <= q < e)
    # records carry (source, index in the source): what the bytes of a dmnd_match would be
    tag = [np.stack([np.full(len(q), g, dtype=np.int64), np.arange(len(q), dtype=np.int64), q.astype(np.int64)], axis=1) for g, q in enumerate(queries)]
    sorted_copy = []
    for g in range(n):
        assert sorted(place[g].tolist()) == list(range(len(queries[g])))          # a permutation
        s = np.empty_like(tag[g])
        s[place[g]] = tag[g]
        sorted_copy.append(s)
    recv_buf = [np.full((int(n_recv[j]), 3), -1, dtype=np.int64) for j in range(n)]
    for g in range(n):
        assert cnt[g].sum() == len(queries[g])
        for j in range(n):
            c = int(cnt[g, j])
            assert send[g, j] == cnt[g, :j].sum() and recv[j, g] == cnt[:g, j].sum()
            recv_buf[j][recv[j, g]:recv[j, g] + c] = sorted_copy[g][send[g, j]:send[g, j] + c]
    seen = set()
    for j in range(n):
        assert n_recv[j] == cnt[:, j].sum()
        b = recv_buf[j]
        assert (b[:, 0] >= 0).all()                                                # every slot written
        for src, idx, q in b.tolist():
            assert owner(q) == j and (src, idx) not in seen
            seen.add((src, idx))
        # inside an owner: sources in rank order, a source's records in their original (query) order -- what dmnd_join_blocks_device's
        # stable sort by query turns into the reference's block order
        keys = b[:, 0] * (1 << 40) + b[:, 1]
        assert (np.diff(keys) > 0).all() if len(keys) > 1 else True
    assert len(seen) == sum(sizes)


def test_a_query_outside_the_range_is_refused():
    lib = hip.load()
    rc, *_ = plan(lib, [np.array([0, 5], dtype=np.uint32), np.array([7], dtype=np.uint32)], 7)
    assert rc != 0
