#!/usr/bin/env python3
"""The in-agent shape of processor_grok: ProcessorRunner hands a plugin ONE event group (~1000 logs) at a time, from
process_thread_count threads that share the plugin instance (core/runner/ProcessorRunner.cpp:138-142).  Here: T host threads,
each calling lc_grok_match_host (values in host memory -> copy in, match on the device, fields back) on its own 1000-line groups
of the configs[2] corpus, the 50-entry example_config Match list.  Prints one JSON line per thread count.

    python tools/grok_inagent_bench.py --threads 1,16 --group 1000 --groups 40
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # a process that hosts a Grok processor (csrc/gpu_runtime.hip); before torch touches the GPU

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="1,16")
    ap.add_argument("--group", type=int, default=1000)
    ap.add_argument("--groups", type=int, default=40, help="groups per thread in the timed region")
    ap.add_argument("--patterns", type=int, default=0)
    args = ap.parse_args()
    for out in measure(args):
        print(json.dumps(out), flush=True)


def measure(args):
    """-> one result dict per thread count (bench.py embeds them under configs[2])"""
    results = []

    from loongcollector_amd import binding
    from loongcollector_amd.grok import Grok, _lib
    from loongcollector_amd.grok_corpus import grok_lines
    from tools.grok_bench import supported_patterns

    if binding.device_count() <= 0:
        raise SystemExit("needs a HIP device: the Grok matcher has no CPU path")
    with open(os.path.join(ROOT, "tests", "golden", "grok_config3.json"), encoding="utf-8") as f:
        cfg = json.load(f)
    supported, _ = supported_patterns(cfg)
    if args.patterns:
        supported = supported[:args.patterns]
    g = Grok(Match=supported, CustomPatterns=cfg["custom_patterns"]).wait_ready()
    L = _lib()
    max_threads = max(int(t) for t in args.threads.split(","))
    n_groups_pool = 8
    values = grok_lines(args.group * n_groups_pool)
    groups = []
    for k in range(n_groups_pool):
        vs = values[k * args.group:(k + 1) * args.group]
        length = np.array([len(v) for v in vs], dtype=np.uint32)
        off = np.zeros(len(vs), dtype=np.uint32)
        off[1:] = np.cumsum(length[:-1], dtype=np.uint64).astype(np.uint32)
        data = np.frombuffer(b"".join(vs) + b"\0" * 16, dtype=np.uint8).copy()
        groups.append((data, off, length, vs))
    # parity gate: one group through the host entry against the oracle (strided)
    from oracle.grok_oracle import GrokOracle
    o = GrokOracle(supported, custom_patterns=cfg["custom_patterns"])
    # the lazy automata learn from the handle's traffic (include/lc_grok.h): the groups go through once or twice behind the trainer first
    for _ in range(3):
        for gr in groups:
            g.match_host(gr[3])
            g.lazy_settle()
    pattern, fields = g.match_host(groups[0][3])
    for i in range(0, args.group, 10):
        res, want = o.process_value(groups[0][3][i])
        if (pattern[i] >= 0) != (res == 0) or fields[i] != want:
            raise SystemExit("PARITY FAILURE: in-agent group, value %d" % i)

    # (the arguments of every call are made once: sixteen Python threads share one interpreter lock, and what a thread does under it
    # between two calls -- numpy's .ctypes objects, byref -- is time the fifteen others wait before they can call in again)
    group_args = [(d.ctypes.data, o.ctypes.data, ln.ctypes.data, len(o)) for d, o, ln, _ in groups]
    match_host, result_free, handle = L.lc_grok_match_host, L.lc_grok_result_free, g._h

    def worker(tid, n_groups, barrier, out):
        pattern = np.empty(args.group, dtype=np.int32)
        pattern_ptr = pattern.ctypes.data
        res = ctypes.c_void_p()
        res_ref = ctypes.byref(res)
        def one(k):
            a = group_args[(tid + k) % n_groups_pool]
            rc = match_host(handle, a[0], a[1], a[2], a[3], pattern_ptr, res_ref)
            if rc != 0:
                raise RuntimeError("lc_grok_match_host rc=%d" % rc)
            result_free(res)
        for k in range(3):
            one(k)
        barrier.wait()
        t0 = time.perf_counter()
        for k in range(n_groups):
            one(k)
        out[tid] = time.perf_counter() - t0
        binding.load().lc_thread_release()

    for t in [int(x) for x in args.threads.split(",")]:
        cs0 = g.combiner_stats()
        barrier = threading.Barrier(t + 1)
        out = [0.0] * t
        threads = [threading.Thread(target=worker, args=(i, args.groups, barrier, out)) for i in range(t)]
        for th in threads:
            th.start()
        barrier.wait()
        t0 = time.perf_counter()
        for th in threads:
            th.join()
        # (the slowest thread's timed loop.  Through round 5 this was the time until the last thread had JOINED -- which includes
        # lc_thread_release: seventeen streams, the plan's events and a dozen device buffers given back per thread, 5-10 ms = 0.3-0.8 ms
        # per group of a 12-20 group run.  Runner threads of an agent live as long as the agent does.)
        del t0
        wall = max(out)
        lines = t * args.groups * args.group
        mean_bytes = float(np.mean([gr[2].sum() for gr in groups])) / args.group
        cs1 = g.combiner_stats()
        nb = max(1, cs1["batches"] - cs0["batches"])
        combiner = {"batches": cs1["batches"] - cs0["batches"], "groups_per_batch": round((cs1["groups"] - cs0["groups"]) / nb, 2),
                    "linger_expired": cs1["linger_expired"] - cs0["linger_expired"],
                    "worker_ms_per_batch": {k: round((cs1["worker_us"][k] - cs0["worker_us"][k]) / nb / 1e3, 3) for k in cs1["worker_us"]}}
        results.append({
            "metric": "Grok lines/s, in-agent shape (%d-line groups through lc_grok_match_host)" % args.group,
            "combiner": combiner,
            "value": round(lines / wall, 1), "unit": "lines/s", "runner_threads": t, "groups_per_thread": args.groups,
            "ms_per_group": round(wall / args.groups * 1e3, 3), "MBps": round(lines * mean_bytes / wall / 1e6, 2),
            "config": {"workload": "configs[2] corpus in %d-line groups, %d Match entries, host memory in, fields out" % (args.group, len(supported)),
                       "parity": "group 0 against the oracle, every 10th value"}})
    return results


if __name__ == "__main__":
    main()
