"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The hot path shards by query (SURVEY.md 8e option 1): every rank extends its own query slice against the
database block resident in its HBM, so there is NO collective on the data path. The only exchange is the
final gather of fixed-size per-query top-k records, ordered as the reference's cross-block merge orders
them (JoinRecord::cmp_evalue: evalue asc, score desc, target oid asc; output/join_blocks.cpp:129-137)."""
import numpy as np
import torch
import torch.distributed as dist

TOPK = 25       # max_target_seqs default, basic/config.h:55


def shard_range(n, world, rank):
    """Contiguous query slice of rank (so that concatenating rank outputs preserves query order)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def topk_records(n_queries, query_idx, evalue, score, target_oid, k=TOPK, presorted=False):
    """Packs per-query top-k (evalue, -score, oid) records into a dense [n_queries, k, 3] float64 tensor
    (+inf padded). Inputs must already be culled to <= k rows per query; rows of a query keep input order.
    presorted: rows are already in (query, evalue, -score, oid) order, as dmnd_extend returns them."""
    rec = np.full((n_queries, k, 3), np.inf)
    if len(query_idx):
        if presorted:
            order = slice(None)
        else:
            order = np.lexsort((target_oid, -np.asarray(score, np.int64), evalue, query_idx))
        q = np.asarray(query_idx)[order]
        start = np.r_[0, np.nonzero(np.diff(q))[0] + 1]
        rank_in_q = np.arange(q.size) - np.repeat(start, np.diff(np.r_[start, q.size]))
        keep = rank_in_q < k
        rec[q[keep], rank_in_q[keep], 0] = np.asarray(evalue)[order][keep]
        rec[q[keep], rank_in_q[keep], 1] = -np.asarray(score, np.float64)[order][keep]
        rec[q[keep], rank_in_q[keep], 2] = np.asarray(target_oid, np.float64)[order][keep]
    return torch.from_numpy(rec)


def gather_records(rec, device):
    """ONE all_gather of the record tensors; returns [world, rows, k, 3] on `device`. Ranks may hold different numbers of query
    rows (shard_range hands out slices that differ by one when the query count is not a multiple of the world size): the
    tensors are padded with +inf rows to the largest count (one small all_gather of the counts), which read as "no record"."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rec.unsqueeze(0)
    world = dist.get_world_size()
    counts = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(counts, torch.tensor([rec.shape[0]], dtype=torch.int64, device=device))
    rows = int(counts.max().item())
    rec_d = rec.to(device)
    if rec.shape[0] < rows:
        pad = torch.full((rows - rec.shape[0],) + tuple(rec.shape[1:]), float("inf"), dtype=rec.dtype, device=device)
        rec_d = torch.cat([rec_d, pad])
    rec_d = rec_d.contiguous()
    out = torch.empty((world * rows,) + tuple(rec.shape[1:]), dtype=rec.dtype, device=device)
    dist.all_gather_into_tensor(out, rec_d)          # concatenation along dim 0 (accepted by RCCL and gloo)
    return out.view((world, rows) + tuple(rec.shape[1:]))


def concat_query_shards(gathered, n_queries):
    """Query sharding: rank r holds the rows of shard_range(n_queries, world, r); returns the [n_queries, k, 3] records in
    query order, dropping the padding rows of the shorter shards."""
    world = gathered.shape[0]
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_queries, world, r)
        parts.append(gathered[r, : hi - lo])
    return torch.cat(parts)


def merge_topk(gathered, k=TOPK):
    """Database-sharded variant (SURVEY.md 8e option 2): every rank saw ALL queries against its own shard;
    merges [world, nq, k, 3] into the global per-query top-k with the reference's ordering."""
    world, nq, kk, _ = gathered.shape
    allrec = gathered.permute(1, 0, 2, 3).reshape(nq, world * kk, 3)
    for key in (2, 1, 0):       # stable sorts, least significant key first
        order = torch.sort(allrec[:, :, key], dim=1, stable=True).indices
        allrec = torch.gather(allrec, 1, order.unsqueeze(-1).expand(-1, -1, 3))
    return allrec[:, :k]


def aligned_queries(gathered):
    return int((gathered[..., 0, 0] < float("inf")).sum().item())


def gather_matches(matches, device, target_base=0):
    """Database sharding (SURVEY.md 8e option 2, BASELINE config C5): this rank extended ALL queries against its own shard of
    the database. Gathers the ranks' match records (variable length: one all_gather of the counts, one of the zero-padded
    byte tensors) and returns their concatenation with `target_base` (ordinal of the shard's first sequence) added to the
    targets -- the input of `hip.join_blocks`, which merges them exactly as the reference joins the blocks of a `-b` run."""
    from . import hip
    rec = np.ascontiguousarray(matches, dtype=hip.MATCH_DTYPE).copy()
    rec["target"] += np.uint32(target_base)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rec
    world = dist.get_world_size()
    n = torch.tensor([rec.size], dtype=torch.int64, device=device)
    counts = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(counts, n)
    counts = counts.cpu().numpy()
    cap = int(counts.max())
    item = hip.MATCH_DTYPE.itemsize
    buf = np.zeros(cap * item, np.uint8)
    buf[: rec.size * item] = rec.view(np.uint8).reshape(-1)
    mine = torch.from_numpy(buf).to(device)
    out = torch.empty(world * cap * item, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(out, mine)
    out = out.cpu().numpy().reshape(world, cap * item)
    parts = [out[r, : int(counts[r]) * item].view(hip.MATCH_DTYPE) for r in range(world)]
    return np.concatenate(parts) if parts else rec


def db_shard_join(matches, device, target_base=0, k=TOPK):
    """gather_matches + the reference's block join: every rank returns the same joined records (grouped by query)."""
    from . import hip
    return hip.join_blocks(gather_matches(matches, device, target_base), k)
