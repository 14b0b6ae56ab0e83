"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

Database sharding (SURVEY.md 8e option 2, what bench.py runs): rank g holds shard g of the database and sees all queries;
there is NO collective on the data path. The one exchange is `query_range_join`: an all-to-all of match records keyed by
query range, rank g merging the queries [g Q/G, (g+1) Q/G) as the reference joins reference blocks
(JoinRecord::cmp_evalue: evalue asc, score desc, target oid asc; output/join_blocks.cpp:129-137), then one gather of the
joined records to rank 0. Query sharding (8e option 1) needs only the ordered gather."""
import time

import numpy as np
import torch
import torch.distributed as dist

TOPK = 25       # max_target_seqs default, basic/config.h:55


def shard_range(n, world, rank):
    """Contiguous query slice of rank (so that concatenating rank outputs preserves query order)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# what the exchanges of this process moved so far (bench.py reports it per step: the SCALE record's proof that the collective ran
# over `world_size` ranks): bytes this rank sent / received in payload collectives, their wall time, the calls
STATS = {"bytes_sent": 0, "bytes_received": 0, "exchange_s": 0.0, "collectives": 0, "transport": None}


def _note(sent, received, seconds, transport):
    STATS["bytes_sent"] += int(sent); STATS["bytes_received"] += int(received); STATS["exchange_s"] += float(seconds)
    STATS["collectives"] += 1; STATS["transport"] = transport


def reset_stats():
    STATS.update(bytes_sent=0, bytes_received=0, exchange_s=0.0, collectives=0, transport=None)


def _a2a_bytes(parts, device):
    """all_to_all of variable-length byte strings: parts[g] (uint8 ndarray) goes to rank g; returns the list of the world's
    contributions to this rank, in rank order. Two collectives: the counts, then the payload (all_to_all_single with split
    sizes: RCCL runs it as grouped point-to-point sends over xGMI, gloo as pairwise exchanges)."""
    world = dist.get_world_size()
    n_in = torch.tensor([int(p.size) for p in parts], dtype=torch.int64, device=device)
    n_out = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_to_all_single(n_out, n_in)
    n_out = [int(x) for x in n_out.cpu().tolist()]
    send = torch.from_numpy(np.concatenate(parts) if sum(p.size for p in parts) else np.zeros(0, np.uint8)).to(device)
    recv = torch.empty(sum(n_out), dtype=torch.uint8, device=device)
    t0 = time.perf_counter()
    dist.all_to_all_single(recv, send, output_split_sizes=n_out, input_split_sizes=[int(p.size) for p in parts])
    recv = recv.cpu().numpy()
    _note(send.numel(), recv.size, time.perf_counter() - t0, "all_to_all_single (%s, through host memory)" % dist.get_backend())
    cuts = np.cumsum([0] + n_out)
    return [recv[cuts[r]:cuts[r + 1]] for r in range(world)]


def _a2a_device(send, in_counts, device):
    """all_to_all of device-resident bytes: `send` (uint8 tensor on `device`) holds the parts for ranks 0..G-1 back to back,
    in_counts[g] bytes each. Returns (recv tensor on the device, per-source byte counts). The counts travel first (one small
    collective whose result the host needs for the split sizes), then the payload, device to device over xGMI."""
    world = dist.get_world_size()
    n_in = torch.tensor([int(x) for x in in_counts], dtype=torch.int64, device=device)
    n_out = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_to_all_single(n_out, n_in)
    n_out = [int(x) for x in n_out.cpu().tolist()]
    recv = torch.empty(sum(n_out), dtype=torch.uint8, device=device)
    t0 = time.perf_counter()
    dist.all_to_all_single(recv, send, output_split_sizes=n_out, input_split_sizes=[int(x) for x in in_counts])
    if recv.is_cuda:
        torch.cuda.current_stream(device).synchronize()
    _note(send.numel(), recv.numel(), time.perf_counter() - t0, "all_to_all_single (%s, device tensors)" % dist.get_backend())
    return recv, n_out


def query_range_join_device(matches, n_queries, device, ctx, k=TOPK, root=0):
    """query_range_join with the records DEVICE-RESIDENT from the first all-to-all to the gather (round 5): one upload of this rank's
    records (ordered by destination on the host, where dmnd_extend leaves them), all_to_all_single on device tensors (RCCL: grouped
    sends over xGMI), the merge of this rank's query range ON THE DEVICE (dmnd_join_blocks_device: radix sorts of a permutation
    + top-k per query), the survivors sent on to `root` from where they lie, ONE download there. `ctx`: a hip.Context on `device`
    (its stream and scratch run the join). One record per (query, target): --max-hsps 1, no range culling (else: query_range_join)."""
    from . import hip
    rec = np.ascontiguousarray(matches, dtype=hip.MATCH_DTYPE)
    world, rank = dist.get_world_size(), dist.get_rank()
    item = hip.MATCH_DTYPE.itemsize
    hi = np.array([shard_range(n_queries, world, g)[1] for g in range(world)], dtype=np.int64)
    dest = np.searchsorted(hi, rec["query"].astype(np.int64), side="right")
    assert dest.size == 0 or dest.max() < world, "a record's query lies outside [0, n_queries)"
    order = np.argsort(dest, kind="stable")
    cuts = np.searchsorted(dest[order], np.arange(world + 1))
    send = torch.from_numpy(rec[order].view(np.uint8).reshape(-1)).to(device)                 # the one upload
    recv, n_from = _a2a_device(send, [(cuts[g + 1] - cuts[g]) * item for g in range(world)], device)
    n = recv.numel() // item
    out = torch.empty(max(n, 1) * item, dtype=torch.uint8, device=device)
    torch.cuda.current_stream(device).synchronize()                                            # the join runs on the context's own stream
    n_kept = ctx.join_blocks_device_ptr(recv.data_ptr(), n, out.data_ptr(), max_target_seqs=k, max_query=max(n_queries - 1, 1)) if n else 0
    mine_dev = out[:n_kept * item]
    full_dev, _ = _a2a_device(mine_dev, [mine_dev.numel() if g == root else 0 for g in range(world)], device)      # every rank's survivors go to root
    mine = mine_dev.cpu().numpy().view(hip.MATCH_DTYPE)
    return mine, (full_dev.cpu().numpy().view(hip.MATCH_DTYPE) if rank == root else None)


def query_range_join(matches, n_queries, device, k=TOPK, root=0, own=False, force_exchange=False, ctx=None):
    """SURVEY.md 8(e).2's exchange for database shards. `matches`: this rank's records (all queries against its own shard(s),
    database-wide target ordinals). Step 1: all-to-all keyed by query range -- the records of queries [g Q/G, (g+1) Q/G)
    (`shard_range`) go to rank g, <= k x 96 B per query and shard. Step 2: rank g merges its 1/G of the queries with
    dmnd_join_blocks (the reference's join_query heap merge by JoinRecord::cmp_evalue + GlobalCulling,
    output/join_blocks.cpp:129-137,180-256). Step 3: the joined records travel once more, to `root`, whose concatenation in
    rank order is in query order. Returns (records of this rank's query range, all records on root | None elsewhere).
    own=True: `matches` is a contiguous record array the caller hands over (it may be reordered in place: no defensive copy).
    force_exchange=True: the collectives run even in a group of one rank (the RCCL path of a 1-GPU box, tests).
    See query_range_join_device for the form that keeps the records in HBM between the collectives."""
    from . import hip
    rec = np.ascontiguousarray(matches, dtype=hip.MATCH_DTYPE)
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force_exchange):
        joined = hip.join_blocks(rec, k, copy=own is False)
        return joined, joined
    world, rank = dist.get_world_size(), dist.get_rank()
    item = hip.MATCH_DTYPE.itemsize
    hi = np.array([shard_range(n_queries, world, g)[1] for g in range(world)], dtype=np.int64)
    dest = np.searchsorted(hi, rec["query"].astype(np.int64), side="right")
    assert dest.size == 0 or dest.max() < world, "a record's query lies outside [0, n_queries)"
    order = np.argsort(dest, kind="stable")                  # block order and query order inside a destination survive
    cuts = np.searchsorted(dest[order], np.arange(world + 1))
    raw = rec[order].view(np.uint8).reshape(-1)
    got = _a2a_bytes([raw[cuts[g] * item:cuts[g + 1] * item] for g in range(world)], device)
    # ctx (a hip.Context): the merge runs on its device (dmnd_join_blocks_device_host) although the exchange went through host
    # memory -- two ranks that share one GPU over gloo (tests/test_gpu_db_shard.py)
    union = np.concatenate(got).view(hip.MATCH_DTYPE)
    mine = ctx.join_blocks_device(union, k) if ctx is not None else hip.join_blocks(union, k, copy=False)
    empty = np.zeros(0, np.uint8)
    out = _a2a_bytes([mine.view(np.uint8).reshape(-1) if g == root else empty for g in range(world)], device)
    return mine, (np.concatenate(out).view(hip.MATCH_DTYPE) if rank == root else None)


def gather_to_root(matches, device, root=0):
    """Query sharding (SURVEY.md 8e option 1): rank r extended the contiguous query slice `shard_range(n, world, r)`, so the
    job's records in query order are the ranks' records concatenated in rank order. One exchange, to `root` only."""
    from . import hip
    rec = np.ascontiguousarray(matches, dtype=hip.MATCH_DTYPE)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rec
    empty = np.zeros(0, np.uint8)
    out = _a2a_bytes([rec.view(np.uint8).reshape(-1) if g == root else empty for g in range(dist.get_world_size())], device)
    return np.concatenate(out).view(hip.MATCH_DTYPE) if dist.get_rank() == root else None
