"""-m gpu: the block join on the device (dmnd_join_blocks_device, csrc/join_device.hip; SURVEY.md 8(e) "final top-k merge") against the
host join (dmnd_join_blocks / dmnd_join_blocks_top = join_query's heap merge + GlobalCulling, pinned on the reference binary's blocked
runs by tests/test_gpu_cli.py and on golden records by tests/test_join_blocks.py): byte-equal record arrays
  * on random records built to collide -- few distinct e-values (incl. 0.0) and scores, so that every tie-break of
    JoinRecord::cmp_evalue / cmp_score decides somewhere --, for several k and --top percentages;
  * on the real records of three database blocks searched on the MI355X;
  * through the device-pointer entry (torch tensors), and through multigpu.query_range_join_device over RCCL with the one rank a
    1-GPU box has (backend "nccl": all_to_all_single on device tensors, the merge on the device, one gather)."""
import os
import numpy as np
import pytest
import torch

from diamond_amd import hip, multigpu, synth, workload

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def _random_records(rng, n_queries, per_query, blocks=4):
    parts = []
    for b in range(blocks):
        q = np.repeat(np.arange(n_queries, dtype=np.uint32), rng.integers(0, per_query + 1, n_queries))
        r = np.zeros(q.size, dtype=hip.MATCH_DTYPE)
        r["query"] = q
        # distinct targets per (query, block): a block's own ordinal range
        r["target"] = (b * 1_000_000 + rng.permutation(1_000_000)[:q.size]).astype(np.uint32)
        score = rng.integers(30, 38, q.size).astype(np.int32)
        r["hsp"]["score"] = score
        r["evalue"] = np.choose(rng.integers(0, 4, q.size), [0.0, 1e-30, 2.5e-7, 3.0])
        r["bit_score"] = 0.37 * score + 3.1
        r["d_begin"] = rng.integers(-50, 50, q.size)
        r["hsp"]["q_end"] = rng.integers(1, 300, q.size)
        parts.append(r)                                        # (a block's list is in query order, as dmnd_extend returns it)
    return np.concatenate(parts)


@pytest.fixture(scope="module")
def ctx():
    c = hip.Context(params=hip.default_params())
    yield c
    c.close()


@pytest.mark.parametrize("k", [1, 5, 25])
def test_device_join_equals_host_join_on_colliding_records(ctx, k):
    rec = _random_records(np.random.default_rng(k), 3000, 30)
    want = hip.join_blocks(rec, k)
    got = ctx.join_blocks_device(rec, k)
    assert len(want) > 1000 and len(got) == len(want)
    assert got.tobytes() == want.tobytes()
    assert np.unique(got["query"], return_counts=True)[1].max() == k


@pytest.mark.parametrize("top", [0.0, 4.0, 60.0])
def test_device_join_top_percent_equals_host_join(ctx, top):
    rec = _random_records(np.random.default_rng(17), 2000, 20)
    want = hip.join_blocks_top(rec, top)
    got = ctx.join_blocks_device(rec, 25, top_percent=top)
    assert len(want) > 500 and got.tobytes() == want.tobytes()


def test_device_join_edge_cases(ctx):
    empty = np.zeros(0, dtype=hip.MATCH_DTYPE)
    assert len(ctx.join_blocks_device(empty, 25)) == 0
    one = _random_records(np.random.default_rng(3), 1, 40, blocks=1)
    assert ctx.join_blocks_device(one, 25).tobytes() == hip.join_blocks(one, 25).tobytes()
    sparse = _random_records(np.random.default_rng(4), 500, 6)
    sparse["query"] *= np.uint32(7919)                         # sparse query ids: the 32-bit key path
    assert ctx.join_blocks_device(sparse, 3).tobytes() == hip.join_blocks(sparse, 3).tobytes()


@pytest.fixture(scope="module")
def block_records():
    db, doff, q, qoff = synth.generate(600, members=10, queries=800, seed=91)
    qd, ql = workload.sequence_set(q, qoff)
    n = len(doff) - 1
    cuts = [0, n // 3, 2 * n // 3, n]
    params = hip.default_params()
    params.db_letters = float(doff[-1])
    c = hip.Context(params=params)
    parts = []
    try:
        c.upload_block(hip.QUERY, qd, ql)
        for a, b in zip(cuts[:-1], cuts[1:]):
            td, tl = workload.sequence_set(db[doff[a]:doff[b]], doff[a:b + 1] - doff[a])
            c.upload_block(hip.TARGET, td, tl)
            m, _ = c.extend(qd, td, c.seed_search(hip.seed_params_fast(threads=4)), threads=4)
            m = m.copy()
            m["target"] += np.uint32(a)
            parts.append(m)
    finally:
        c.close()
    return np.concatenate(parts), len(ql) - 1


def test_device_join_on_real_block_records(ctx, block_records):
    rec, nq = block_records
    for k in (25, 3):
        want = hip.join_blocks(rec, k)
        assert len(want) > 300, len(want)
        assert ctx.join_blocks_device(rec, k).tobytes() == want.tobytes()
    want = hip.join_blocks(rec, 25)
    # device pointers: records and result as torch tensors on the context's device
    src = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).cuda()
    out = torch.empty_like(src)
    torch.cuda.synchronize()
    n = ctx.join_blocks_device_ptr(src.data_ptr(), len(rec), out.data_ptr(), max_target_seqs=25, max_query=nq - 1)
    assert n == len(want) and out[:n * hip.MATCH_DTYPE.itemsize].cpu().numpy().tobytes() == want.tobytes()


def test_query_range_join_device_over_rccl_world_size_1():
    """multigpu.query_range_join_device with the one rank a 1-GPU box has: backend "nccl" (= RCCL), the records uploaded once,
    all_to_all_single on device tensors, the merge on the device, the gather from device memory. In a child process so that the
    process group does not outlive the test."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from diamond_amd import hip, multigpu
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", RANK="0", WORLD_SIZE="1")
rng = np.random.default_rng(5)
rec = np.zeros(60000, hip.MATCH_DTYPE)
rec["query"] = np.sort(rng.integers(0, 900, len(rec)))
rec["target"] = rng.permutation(1 << 20)[:len(rec)]
rec["evalue"] = np.choose(rng.integers(0, 3, len(rec)), [0.0, 1e-12, 0.5])
rec["hsp"]["score"] = rng.integers(30, 40, len(rec))
rec["bit_score"] = rec["hsp"]["score"] * 0.38
want = hip.join_blocks(rec.copy(), 25)
device = torch.device("cuda:0")
torch.cuda.set_device(device)
ctx = hip.Context(params=hip.default_params())
dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
mine, full = multigpu.query_range_join_device(rec, 900, device, ctx, k=25)
dist.barrier()
dist.destroy_process_group()
ctx.close()
assert len(want) > 10000 and mine.tobytes() == want.tobytes() and full.tobytes() == want.tobytes(), (len(mine), len(want))
print("RCCL_DEVICE_JOIN_OK", len(want))
''' % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_DEVICE_JOIN_OK" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


def test_join_ranks_one_process_several_contexts(ctx, block_records):
    """dmnd_join_ranks (csrc/rank_join.hip): what `diamond-hip --gpus N` calls per query block. One context: the exchange is RCCL's
    (ncclCommInitAll for the device, ncclSend / ncclRecv to itself inside one group), the merge is the device's. Three contexts on
    the box's one GPU: owners by query range, device-to-device copies in RCCL's place (it refuses several ranks on a device), three
    merges, the concatenation in query order. Both equal the host join of the same records, with -k and with --top."""
    rec, nq = block_records
    want = hip.join_blocks(rec, 25)
    got, transport = hip.join_ranks([ctx], [rec], nq, 25)
    assert transport == 1 and got.tobytes() == want.tobytes()
    others = [hip.Context(params=hip.default_params()) for _ in range(2)]
    try:
        thirds = np.array_split(np.arange(len(rec)), 3)
        parts = [rec[i] for i in thirds]                       # any split of the records over the sources gives the same join
        got, transport = hip.join_ranks([ctx] + others, parts, nq, 25)
        assert transport == 2 and got.tobytes() == want.tobytes()
        got, _ = hip.join_ranks([ctx] + others, [parts[0], parts[1][:0], np.concatenate(parts[1:])], nq, 4)      # a source without records
        assert got.tobytes() == hip.join_blocks(rec, 4).tobytes()
        got, _ = hip.join_ranks([ctx] + others, parts, nq, 25, top_percent=15.0)
        assert got.tobytes() == hip.join_blocks_top(rec, 15.0).tobytes()
    finally:
        for c in others:
            c.close()
