"""Multi-GPU plumbing of the line-sharded parse path.

Lines (and event groups) are independent, so N GPUs never exchange data: every rank owns a contiguous slab of
lines and its own device tables.  The only collective is the one that lets rank 0 report the job: the MAX over ranks
of the elapsed time and the SUM of the per-rank counters (bytes, lines, matched).  On GPUs this runs over RCCL
(backend "nccl"); the same code runs over gloo on CPU, which is how tests/test_shard_gloo.py covers it.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous, balanced slab [lo, hi) of n_items for this rank; slabs tile [0, n_items) exactly."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def reduce_job(elapsed_s, counters, device="cpu"):
    """-> (max elapsed over ranks, {name: sum over ranks}).  No-op when torch.distributed is not initialised."""
    names = sorted(counters)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(elapsed_s), {k: int(counters[k]) for k in names}
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    c = torch.tensor([int(counters[k]) for k in names], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), {k: int(v) for k, v in zip(names, c.tolist())}
