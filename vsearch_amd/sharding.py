"""Multi-GPU layout of the path: every (query, target) pair is independent, so ranks
take contiguous blocks of QUERIES (the DB is replicated in each GPU's HBM), align their own pairs with
no data-path collective, and gather the fixed-size hit records once at the end (RCCL on GPUs; the same
code runs over gloo on CPU tensors in the tests)."""
import numpy as np


def shard_queries(n_queries, world, rank):
    """contiguous, balanced block of query indices for `rank`: [lo, hi)"""
    base, rem = divmod(int(n_queries), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard_pairs(qidx, tidx, n_queries, world, rank):
    """pairs whose query falls into this rank's block; returns (local_qidx, tidx, global_pair_index)"""
    qidx = np.asarray(qidx)
    tidx = np.asarray(tidx)
    lo, hi = shard_queries(n_queries, world, rank)
    sel = np.nonzero((qidx >= lo) & (qidx < hi))[0]
    return (qidx[sel] - lo).astype(np.uint32), tidx[sel].astype(np.uint32), sel.astype(np.int64), (lo, hi)


def gather_hits(local_records, global_index, n_total, dist=None, device=None):
    """All-gather variable-length blocks of 24-byte hit records and scatter them into global pair order.

    local_records: uint8 tensor (n_local, 24); global_index: int64 tensor (n_local,).  With dist=None the
    call is the single-rank identity.  Returns a (n_total, 24) uint8 tensor on every rank."""
    import torch
    if dist is None or dist.get_world_size() == 1:
        out = torch.zeros((n_total, 24), dtype=torch.uint8, device=local_records.device)
        out[global_index] = local_records
        return out
    world = dist.get_world_size()
    dev = local_records.device
    n_local = torch.tensor([local_records.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, n_local)
    counts = [int(c.item()) for c in counts]
    cap = max(counts + [1])
    pad_r = torch.zeros((cap, 24), dtype=torch.uint8, device=dev)
    pad_i = torch.full((cap,), -1, dtype=torch.int64, device=dev)
    pad_r[:local_records.shape[0]] = local_records
    pad_i[:local_records.shape[0]] = global_index
    all_r = [torch.zeros_like(pad_r) for _ in range(world)]
    all_i = [torch.zeros_like(pad_i) for _ in range(world)]
    dist.all_gather(all_r, pad_r)
    dist.all_gather(all_i, pad_i)
    out = torch.zeros((n_total, 24), dtype=torch.uint8, device=dev)
    for r in range(world):
        n = counts[r]
        out[all_i[r][:n]] = all_r[r][:n]
    return out
