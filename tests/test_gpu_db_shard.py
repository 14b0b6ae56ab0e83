"""-m gpu: database sharding (SURVEY.md 8e option 2 / BASELINE config C5) end to end with two ranks on ONE GPU (gloo for the
record exchange, as bench.py's DMND_BENCH_SHARE_GPU hook does): every rank searches ALL queries against its own shard on
the MI355X, the ranks exchange their match records by query range (multigpu.query_range_join) and join them as the reference joins the blocks of a `-b` run. Must equal
the same two blocks processed one after the other in a single process (whose text tests/test_gpu_cli.py pins against the
reference binary run with the same block boundaries)."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _blocks():
    from diamond_amd import synth, workload
    db, doff, q, qoff = synth.generate(400, members=10, queries=500, seed=77)
    qd, ql = workload.sequence_set(q, qoff)
    half = (len(doff) - 1) // 2
    shards = []
    for a, b in ((0, half), (half, len(doff) - 1)):
        td, tl = workload.sequence_set(db[doff[a]:doff[b]], doff[a:b + 1] - doff[a])
        shards.append((a, td, tl))
    return qd, ql, shards, float(doff[-1])


def _search(ctx, hip, qd, ql, td, tl, sp):
    ctx.upload_block(hip.QUERY, qd, ql)
    ctx.upload_block(hip.TARGET, td, tl)
    hits = ctx.seed_search(sp)
    m, _ = ctx.extend(qd, td, hits, threads=4)
    return m


def _rank(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    from diamond_amd import hip, multigpu
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    qd, ql, shards, letters = _blocks()
    params = hip.default_params()
    params.db_letters = letters                                   # e-values against the WHOLE database on every rank
    ctx = hip.Context(device=0, params=params)
    try:
        base, td, tl = shards[rank]
        m = _search(ctx, hip, qd, ql, td, tl, hip.seed_params_fast(threads=4))
        m = m.copy()
        m["target"] += np.uint32(base)
        # rank 1 merges its query range on the device (dmnd_join_blocks_device), rank 0 on the host: the same records either way
        _, joined = multigpu.query_range_join(m, len(ql) - 1, torch.device("cpu"), k=25, ctx=ctx if rank == 1 else None)
    finally:
        ctx.close()
    ret[rank] = None if joined is None else joined.tobytes()
    dist.destroy_process_group()


def test_two_rank_database_shards_equal_sequential_blocks():
    assert torch.cuda.is_available()
    sys.path.insert(0, ROOT)
    from diamond_amd import hip
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rank, args=(world, 29541, ret), nprocs=world, join=True)
    qd, ql, shards, letters = _blocks()
    params = hip.default_params()
    params.db_letters = letters
    ctx = hip.Context(params=params)
    try:
        parts = []
        for base, td, tl in shards:
            m = _search(ctx, hip, qd, ql, td, tl, hip.seed_params_fast(threads=4)).copy()
            m["target"] += np.uint32(base)
            parts.append(m)
    finally:
        ctx.close()
    want = hip.join_blocks(np.concatenate(parts), 25)
    assert len(want) > 300 and len(set(want["target"].tolist())) > 300
    assert ret[1] is None and ret[0] == want.tobytes()


def test_query_range_join_over_rccl_world_size_1(tmp_path):
    """The RCCL branch of multigpu.query_range_join on the box's one GPU: backend "nccl" (= RCCL on ROCm), a process group of one
    rank, device tensors, all_to_all_single with split sizes -- forced to run although a single rank has nothing to exchange --
    and the same records as the local join (and as a gloo group doing the same). Runs in a child process so that the process
    group does not outlive the test."""
    import subprocess
    import sys
    code = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from diamond_amd import hip, multigpu
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
rng = np.random.default_rng(3)
rec = np.zeros(4000, hip.MATCH_DTYPE)
rec["query"] = np.sort(rng.integers(0, 300, len(rec)))
rec["target"] = rng.integers(0, 100000, len(rec))
rec["evalue"] = 10.0 ** rng.integers(-80, -3, len(rec))
rec["hsp"]["score"] = rng.integers(30, 900, len(rec))
rec["bit_score"] = rec["hsp"]["score"] * 0.38
local = hip.join_blocks(rec.copy(), 25)
device = torch.device("cuda:0")
torch.cuda.set_device(device)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
mine, full = multigpu.query_range_join(rec.copy(), 300, device, force_exchange=True)
dist.barrier()
dist.destroy_process_group()
assert len(local) > 1000 and np.array_equal(mine, local) and np.array_equal(full, local), (len(mine), len(local))
dist.init_process_group("gloo", rank=0, world_size=1)
mine2, full2 = multigpu.query_range_join(rec.copy(), 300, torch.device("cpu"), force_exchange=True)
dist.destroy_process_group()
assert np.array_equal(mine2, local) and np.array_equal(full2, local)
print("RCCL_OK", len(local))
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
