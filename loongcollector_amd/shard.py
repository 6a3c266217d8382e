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


# ---- BASELINE configs[4]: a corpus fed from host memory in slabs ---------------------------------------------------------
def deal_slabs(n_slabs, rank, world):
    """The slabs of a corpus dealt round-robin to the ranks: rank r owns slabs r, r + world, ...
    Every slab belongs to exactly one rank."""
    return list(range(rank, n_slabs, world))


def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_node(pci_bus_id, sysfs="/sys/bus/pci/devices"):
    """NUMA node of the GPU with this PCI address ("0000:c1:00.0"), or -1 when the kernel does not say."""
    import os
    for name in (pci_bus_id.lower(), pci_bus_id.upper()):
        path = os.path.join(sysfs, name, "numa_node")
        if os.path.exists(path):
            try:
                with open(path) as f:
                    return int(f.read().strip())
            except (OSError, ValueError):
                return -1
    return -1


def place_rank(local_rank, sysfs_pci="/sys/bus/pci/devices", sysfs_node="/sys/devices/system/node", apply=True):
    """Pin the calling process to the CPUs of the NUMA node its GPU hangs off, BEFORE it allocates pinned staging memory (first
    touch then puts the pages next to the GPU's PCIe root): with N GPUs fed from one host, which lanes and which memory
    controller a slab crosses decides the scaling curve, not the kernels.  -> {"pci": ..., "numa_node": ..., "cpus": n, "pinned":
    bool} for the bench line's per_gpu table.  Nothing here is fatal: a box that does not say stays unpinned."""
    import os
    info = {"pci": None, "numa_node": -1, "cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 0,
            "pinned": False}
    try:
        import torch
        if torch.cuda.is_available():
            p = torch.cuda.get_device_properties(local_rank)
            if hasattr(p, "pci_bus_id"):
                info["pci"] = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, getattr(p, "pci_device_id", 0))
    except Exception:  # noqa: BLE001 -- placement is best effort
        pass
    if info["pci"]:
        info["numa_node"] = gpu_numa_node(info["pci"], sysfs_pci)
    if info["numa_node"] >= 0 and hasattr(os, "sched_setaffinity"):
        path = os.path.join(sysfs_node, "node%d" % info["numa_node"], "cpulist")
        try:
            with open(path) as f:
                cpus = [c for c in _parse_cpulist(f.read()) if c in os.sched_getaffinity(0)]
            if cpus and apply:
                os.sched_setaffinity(0, cpus)
                info["pinned"] = True
                info["cpus"] = len(cpus)
        except OSError:
            pass
    return info


def run_slab_job(n_slabs, rank, world, feed, drain, in_flight=3):
    """Drive one rank's share of a slab job: feed(slab_index, buffer_index) queues a slab (returns a ticket), drain(ticket) waits
    for it and books its counters; at most `in_flight` slabs are queued ahead.  -> the slab indices this rank ran, in order."""
    mine = deal_slabs(n_slabs, rank, world)
    queue = []
    for i, s in enumerate(mine):
        if len(queue) == in_flight:
            drain(queue.pop(0))
        queue.append(feed(s, i % in_flight))
    while queue:
        drain(queue.pop(0))
    return mine
