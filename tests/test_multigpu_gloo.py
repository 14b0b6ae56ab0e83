"""world_size-2 CPU test (gloo) of the N>1 path: query sharding + the single all_gather of per-query top-k
records, and the database-sharded merge ordering (JoinRecord::cmp_evalue, output/join_blocks.cpp:129-137)."""
import os
import sys
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _records(seed, nq, n):
    rng = np.random.default_rng(seed)
    q = np.sort(rng.integers(0, nq, n))
    ev = rng.choice([1e-30, 1e-10, 1e-5, 2e-5], n)          # ties on purpose
    sc = rng.integers(40, 60, n)
    oid = rng.permutation(n) + seed * 100000
    return q, ev, sc, oid


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    from diamond_amd import multigpu
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nq = 50
    q, ev, sc, oid = _records(rank + 1, nq, 400)
    rec = multigpu.topk_records(nq, q, ev, sc, oid)
    g = multigpu.gather_records(rec, torch.device("cpu"))
    merged = multigpu.merge_topk(g)
    lo, hi = multigpu.shard_range(101, world, rank)
    ret[rank] = (g.numpy().copy(), merged.numpy().copy(), (lo, hi))
    dist.destroy_process_group()


def test_two_rank_gather_and_merge():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, 29517, ret), nprocs=world, join=True)
    g0, m0, r0 = ret[0]
    g1, m1, r1 = ret[1]
    assert np.array_equal(g0, g1) and np.array_equal(m0, m1)          # every rank holds the same gathered view
    assert r0 == (0, 51) and r1 == (51, 101)                            # contiguous, covering, ordered slices
    # reference ordering, brute force
    nq = 50
    rows = []
    for rank in range(world):
        q, ev, sc, oid = _records(rank + 1, nq, 400)
        rows += list(zip(q, ev, -sc.astype(float), oid.astype(float)))
    for qi in range(nq):
        mine = sorted((r[1:] for r in rows if r[0] == qi))
        # each rank first culls to its own top-25, then the merge keeps the global top-25 of those
        per_rank = []
        for rank in range(world):
            q, ev, sc, oid = _records(rank + 1, nq, 400)
            rr = sorted((e, -float(s), float(o)) for qq, e, s, o in zip(q, ev, sc, oid) if qq == qi)[:25]
            per_rank += rr
        want = sorted(per_rank)[:25]
        got = [tuple(x) for x in m0[qi] if np.isfinite(x[0])]
        assert got == want
        assert len(mine) >= len(got)


def _shard_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from diamond_amd import multigpu
    from test_join_blocks import _block_records
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blocks = _block_records(np.random.default_rng(5), 40, world, 300)          # every rank draws the same set and keeps its own shard
    mine = blocks[rank].copy()
    mine["target"] -= np.uint32(rank * 300)                                     # shard-local target ids, as dmnd_extend returns them
    joined = multigpu.db_shard_join(mine, torch.device("cpu"), target_base=rank * 300, k=25)
    ret[rank] = joined.tobytes()
    dist.destroy_process_group()


def test_two_rank_database_shard_join_equals_block_join():
    """Database sharding (SURVEY.md 8e option 2): two ranks, one shard each, all queries; the gathered + joined records equal
    the sequential join of the same two blocks (dmnd_join_blocks, pinned on the reference's heap merge in test_join_blocks)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from diamond_amd import hip
    from test_join_blocks import _block_records
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_shard_worker, args=(world, 29519, ret), nprocs=world, join=True)
    blocks = _block_records(np.random.default_rng(5), 40, world, 300)
    want = hip.join_blocks(np.concatenate(blocks), 25)
    assert ret[0] == ret[1] == want.tobytes()
    assert len(want) > 500


def _uneven_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    from diamond_amd import multigpu
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nq = 101                                                   # 51 + 50 queries
    lo, hi = multigpu.shard_range(nq, world, rank)
    q, ev, sc, oid = _records(7, nq, 900)                      # every rank draws the same records and keeps its own queries
    keep = (q >= lo) & (q < hi)
    rec = multigpu.topk_records(hi - lo, q[keep] - lo, ev[keep], sc[keep], oid[keep])
    g = multigpu.gather_records(rec, torch.device("cpu"))
    ret[rank] = multigpu.concat_query_shards(g, nq).numpy().copy()
    dist.destroy_process_group()


def test_query_shards_of_unequal_size_gather():
    """101 queries over 2 ranks: the record tensors have 51 and 50 rows; the gather pads, the concatenation trims."""
    sys.path.insert(0, ROOT)
    from diamond_amd import multigpu
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_uneven_worker, args=(world, 29523, ret), nprocs=world, join=True)
    q, ev, sc, oid = _records(7, 101, 900)
    want = multigpu.topk_records(101, q, ev, sc, oid).numpy()
    assert ret[0].shape == (101, multigpu.TOPK, 3)
    assert np.array_equal(ret[0], ret[1]) and np.array_equal(ret[0], want)
