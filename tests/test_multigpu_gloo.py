"""world_size-2 / 3 CPU tests (gloo) of the N > 1 path:
  * database shards: `multigpu.query_range_join` -- all-to-all of match records keyed by query range, rank g joining the
    queries [g Q/G, (g+1) Q/G) with dmnd_join_blocks, one gather to rank 0 -- equals the sequential join of the same blocks
    (dmnd_join_blocks itself is pinned on the reference's heap merge, JoinRecord::cmp_evalue output/join_blocks.cpp:129-137,
    in test_join_blocks); uneven query ranges, several blocks per rank, a rank without records, a query count below the world size;
  * query shards: the ordered gather to rank 0;
  * bench.py's launcher: `--gpus N` without torchrun re-executes itself under torch.distributed.run with N ranks, and a rank
    count that differs from --gpus is a hard failure."""
import os
import subprocess
import sys
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _init(rank, world, port):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _shard_worker(rank, world, port, case, ret):
    _init(rank, world, port)
    from diamond_amd import multigpu
    from test_join_blocks import _block_records, _expand_runs
    nq, n_blocks, tpb, k = case
    blocks = _block_records(np.random.default_rng(5), nq, n_blocks, tpb)        # every rank draws the same set ...
    if case == CASES[4]:
        blocks = _expand_runs(blocks, np.random.default_rng(6))                   # --max-hsps: runs of records per (query, target)
    mine = [blocks[b] for b in range(rank, n_blocks, world)]                      # ... and keeps blocks rank, rank + world, ...
    if case == CASES[2] and rank == 1:
        mine = []                                                                 # a rank whose shard produced no alignment
    rec = np.concatenate(mine) if mine else np.zeros(0, blocks[0].dtype)
    part, full = multigpu.query_range_join(rec, nq, torch.device("cpu"), k=k)
    lo, hi = multigpu.shard_range(nq, world, rank)
    assert part.size == 0 or (part["query"].min() >= lo and part["query"].max() < hi)
    assert (full is None) == (rank != 0)
    ret[rank] = (part.tobytes(), None if full is None else full.tobytes())
    dist.destroy_process_group()


#          queries, blocks, targets per block, k
CASES = [(41, 2, 300, 25), (40, 5, 200, 25), (33, 2, 300, 25), (1, 2, 300, 3), (37, 4, 250, 5)]


def _run_case(ci, world, port):
    from diamond_amd import hip
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_join_blocks import _block_records, _expand_runs
    case = CASES[ci]
    nq, n_blocks, tpb, k = case
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_shard_worker, args=(world, port, case, ret), nprocs=world, join=True)
    blocks = _block_records(np.random.default_rng(5), nq, n_blocks, tpb)
    if ci == 4:
        blocks = _expand_runs(blocks, np.random.default_rng(6))
    if ci == 2:
        blocks = [b for i, b in enumerate(blocks) if i % world != 1]
    want = hip.join_blocks(np.concatenate(blocks), k)
    assert ret[0][1] == want.tobytes()                                  # rank 0 holds the whole job's records, in query order
    assert b"".join(ret[r][0] for r in range(world)) == want.tobytes()  # and the ranks' query ranges tile it
    return want


def test_two_rank_query_range_join_equals_block_join():
    want = _run_case(0, 2, 29519)                                        # 41 queries: ranges of 21 and 20
    assert len(want) > 500


def test_three_ranks_five_blocks():
    _run_case(1, 3, 29521)                                               # ranks hold 2, 2 and 1 blocks


def test_rank_without_records_and_fewer_queries_than_ranks():
    _run_case(2, 2, 29523)
    _run_case(3, 2, 29525)                                               # one query: rank 1's range is empty


def test_hsp_runs_stay_together_over_the_ranks():
    """--max-hsps: the records of a (query, target) pair travel as one run through the all-to-all and the per-rank join."""
    want = _run_case(4, 2, 29529)
    assert (want["hsp"]["q_begin"] > 0).sum() > 100


def _gather_worker(rank, world, port, ret):
    _init(rank, world, port)
    from diamond_amd import hip, multigpu
    from test_join_blocks import _block_records
    nq = 101                                                             # 51 + 50 queries
    rec = hip.join_blocks(_block_records(np.random.default_rng(9), nq, 1, 400)[0], 25)
    lo, hi = multigpu.shard_range(nq, world, rank)
    mine = rec[(rec["query"] >= lo) & (rec["query"] < hi)]
    full = multigpu.gather_to_root(mine, torch.device("cpu"))
    ret[rank] = None if full is None else full.tobytes()
    dist.destroy_process_group()


def test_query_shards_of_unequal_size_gather_in_query_order():
    from diamond_amd import hip, multigpu
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_join_blocks import _block_records
    assert multigpu.shard_range(101, 2, 0) == (0, 51) and multigpu.shard_range(101, 2, 1) == (51, 101)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gather_worker, args=(2, 29527, ret), nprocs=2, join=True)
    want = hip.join_blocks(_block_records(np.random.default_rng(9), 101, 1, 400)[0], 25)
    assert ret[1] is None and ret[0] == want.tobytes()


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` (no torchrun): the launcher half of bench.py runs before anything touches the GPU, so it is
    testable here -- DMND_BENCH_LAUNCH_ONLY makes every spawned rank print its rendezvous environment and exit."""
    env = dict(os.environ, DMND_BENCH_LAUNCH_ONLY="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = sorted(l for l in r.stdout.splitlines() if l.startswith("launch-only"))
    assert lines == ["launch-only rank 0 of 2 local 0 gpus 2", "launch-only rank 1 of 2 local 1 gpus 2"], r.stdout + r.stderr[-2000:]
    # a rank count that differs from --gpus is refused, not silently run on fewer GPUs
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env2, cwd=ROOT)
    assert r.returncode != 0 and "--gpus 2" in r.stderr and "WORLD_SIZE=1" in r.stderr
