"""World-size-2 gloo run of the multi-GPU plumbing (loongcollector_amd/shard.py) on CPU: slabs tile the corpus
exactly and the job reduction is MAX(elapsed) / SUM(counters).  The data path itself has no collective."""
import os
import socket

import torch.multiprocessing as mp

from loongcollector_amd.shard import shard_range


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    import torch.distributed as dist
    from loongcollector_amd import corpus
    from loongcollector_amd.shard import reduce_job, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_total = 1001
    lo, hi = shard_range(n_total, rank, world)
    data, off, length = corpus.apache_batch(n_total, "A", pool_lines=64)
    my_bytes = int(length[lo:hi].sum())
    elapsed, total = reduce_job(0.5 + rank, {"bytes": my_bytes, "lines": hi - lo})
    from loongcollector_amd.shard import gather_job, job_totals
    table = gather_job({"bytes": my_bytes, "lines": hi - lo, "elapsed_us": 500000 + 1000000 * rank, "kernel_us": 7 + rank})
    out[rank] = (lo, hi, elapsed, total["bytes"], total["lines"], int(length.sum()))
    out["table%d" % rank] = (table, job_totals(table))
    dist.destroy_process_group()


def test_two_rank_line_shard_and_job_reduction():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    (lo0, hi0, e0, b0, l0, all_bytes), (lo1, hi1, e1, b1, l1, _) = out[0], out[1]
    assert (lo0, hi0, lo1, hi1) == (0, 501, 501, 1001)        # slabs tile the corpus, no overlap, no gap
    assert e0 == e1 == 1.5                                   # MAX over ranks
    assert b0 == b1 == all_bytes and l0 == l1 == 1001        # SUM over ranks == whole job
    # the all-gathered per-GPU table (what bench.py prints as per_gpu): every rank holds both rows, in rank order
    (t0, tot0), (t1, tot1) = out["table0"], out["table1"]
    assert t0 == t1 and len(t0) == 2
    assert [g["lines"] for g in t0] == [501, 500] and [g["kernel_us"] for g in t0] == [7, 8]
    assert t0[0]["bytes"] + t0[1]["bytes"] == all_bytes
    assert tot0 == tot1 and tot0["elapsed_us"] == 1500000 and tot0["lines"] == 1001 and tot0["bytes"] == all_bytes


def test_shard_range_tiles_for_any_world_size():
    for n in (0, 1, 7, 64, 1001):
        for world in (1, 2, 3, 8):
            ranges = [shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in ranges]
            assert max(sizes) - min(sizes) <= 1


def _slab_worker(rank, world, port, out):
    import torch.distributed as dist
    from loongcollector_amd.shard import gather_job, job_totals, run_slab_job
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_slabs, in_flight = 11, 3
    counters = {"slabs": 0, "bytes": 0, "max_queued": 0, "elapsed_us": 1000 * (rank + 1)}
    queued = []

    def feed(s, buf):
        assert 0 <= buf < in_flight and buf not in [b for _, b in queued]      # a buffer is never reused while its slab is in flight
        queued.append((s, buf))
        counters["max_queued"] = max(counters["max_queued"], len(queued))
        return (s, buf)

    def drain(ticket):
        assert queued.pop(0) == ticket                                          # slabs leave in the order they were fed
        counters["slabs"] += 1
        counters["bytes"] += 100 + ticket[0]

    mine = run_slab_job(n_slabs, rank, world, feed, drain, in_flight)
    table = gather_job(counters)
    out[rank] = (mine, table, job_totals(table))
    dist.destroy_process_group()


def test_two_rank_slab_dealing_and_gathered_counters():
    """BASELINE configs[4] (bench.py --config 5): slabs dealt round-robin to the ranks, a bounded number in flight per rank, the
    per-GPU counters all-gathered -- every slab exactly once, every rank sees the whole table."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_slab_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    (m0, t0, tot0), (m1, t1, tot1) = out[0], out[1]
    assert m0 == [0, 2, 4, 6, 8, 10] and m1 == [1, 3, 5, 7, 9]
    assert sorted(m0 + m1) == list(range(11))                                   # every slab dealt exactly once
    assert t0 == t1 and [g["slabs"] for g in t0] == [6, 5]
    assert all(g["max_queued"] == 3 for g in t0)
    assert tot0 == tot1 and tot0["slabs"] == 11 and tot0["bytes"] == 11 * 100 + sum(range(11)) and tot0["elapsed_us"] == 2000


def test_numa_placement_reads_sysfs(tmp_path):
    """place_rank's pieces on a fake sysfs tree: the GPU's numa_node file, the node's cpulist"""
    from loongcollector_amd.shard import _parse_cpulist, deal_slabs, gpu_numa_node
    pci = tmp_path / "pci" / "0000:c1:00.0"
    pci.mkdir(parents=True)
    (pci / "numa_node").write_text("3\n")
    assert gpu_numa_node("0000:C1:00.0", str(tmp_path / "pci")) == 3
    assert gpu_numa_node("0000:99:00.0", str(tmp_path / "pci")) == -1
    assert _parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert [deal_slabs(7, r, 3) for r in range(3)] == [[0, 3, 6], [1, 4], [2, 5]]
