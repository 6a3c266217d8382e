"""Multi-GPU plumbing of the line-sharded parse path.

Lines (and event groups) are independent, so N GPUs never exchange data: every rank owns a contiguous slab of
lines and its own device tables.  The only collective is the one that lets rank 0 report the job: ONE all-gather of the
per-GPU counter struct (bytes, lines, matched, elapsed, kernel time ...), from which the aggregate (MAX of the elapsed times,
SUM of the counters) and the per-GPU scaling table are both derived.  On GPUs this runs over RCCL (backend "nccl"); the same
code runs over gloo on CPU, which is how tests/test_shard_gloo.py covers it.
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


def gather_job(counters, device="cpu"):
    """One all-gather of this rank's counters -> list over ranks of {name: value} (every rank gets the whole table).
    Counters are integers (use microseconds for times).  Without an initialised process group: [counters]."""
    names = sorted(counters)
    mine = {k: int(counters[k]) for k in names}
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [mine]
    world = dist.get_world_size()
    c = torch.tensor([mine[k] for k in names], dtype=torch.int64, device=device)
    rows = [torch.empty_like(c) for _ in range(world)]
    dist.all_gather(rows, c)
    return [{k: int(v) for k, v in zip(names, row.tolist())} for row in rows]


def job_totals(per_gpu, max_keys=("elapsed_us",)):
    """Aggregate of a gather_job() table: MAX over ranks for the keys in max_keys, SUM for the others."""
    return {k: (max if k in max_keys else sum)(g[k] for g in per_gpu) for k in per_gpu[0]}
