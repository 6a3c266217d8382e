#!/usr/bin/env python3
"""bench.py -- headline benchmark: MB/s of log bytes parsed per MI355X (512 B Apache-combined lines, 10-field
regex, bit-exact capture offsets), with the HBM-roofline fraction and the host-CPU baseline next to it.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (lc_regex_match_device: the replacement for the per-event
BoostRegexMatch loop, core/plugin/processor/ProcessorParseRegexNative.cpp:115-124,194) over one batch of
1 Mi synthetic lines already resident in HBM.  Multi-GPU is an embarrassingly parallel line shard: each rank owns
its own batch, there is no data-path collective; only the elapsed time is max-reduced (weak scaling).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--lines", type=int, default=1 << 20, help="lines per batch per GPU")
    ap.add_argument("--line-bytes", type=int, default=512)
    ap.add_argument("--regex", choices=["A", "B"], default="A", help="A: 10-group doc regex, B: 11-group benchmark regex")
    ap.add_argument("--engine", choices=["auto", "tdfa", "nfa"], default="auto")
    ap.add_argument("--cpu-sample-lines", type=int, default=1 << 20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from loongcollector_amd import binding, corpus

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the parse engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    pattern = corpus.REGEX_A if args.regex == "A" else corpus.REGEX_B
    engine = {"auto": binding.LC_ENGINE_AUTO, "tdfa": binding.LC_ENGINE_TDFA, "nfa": binding.LC_ENGINE_NFA}[args.engine]
    rx = binding.GpuRegex(pattern, engine=engine)
    G = rx.groups
    info = rx.info()

    n = args.lines
    data, off, length = corpus.apache_batch(n, args.regex, args.line_bytes, seed=corpus.SEED + 1000 * rank)
    parsed_bytes_per_step = int(length.sum())
    d_data = torch.from_numpy(data).to(dev)
    d_off = torch.from_numpy(off.view(np.int32)).to(dev)
    d_caps = torch.empty((n, 2 * G), dtype=torch.int32, device=dev)
    d_status = torch.empty((n,), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream()

    # one step == one lc_regex_match_device_engine() call; arguments are marshalled once so that the timed loop is
    # the C-ABI call itself and not Python argument conversion
    import ctypes
    L = binding.load()
    call_args = (rx.handle, ctypes.c_int(binding.LC_ENGINE_AUTO), ctypes.c_void_p(d_data.data_ptr()),
                 ctypes.c_void_p(d_off.data_ptr()), ctypes.c_void_p(None), ctypes.c_uint32(1), ctypes.c_uint32(n),
                 ctypes.c_uint32(G), ctypes.c_void_p(d_caps.data_ptr()), ctypes.c_void_p(d_status.data_ptr()),
                 ctypes.c_void_p(stream.cuda_stream))
    match_fn = L.lc_regex_match_device_engine

    def step():
        rc = match_fn(*call_args)
        if rc != 0:
            raise SystemExit("lc_regex_match_device failed rc=%d: %s" % (rc, L.lc_last_error()))

    step()  # one untimed pass to produce the capture table the parity gate below checks
    torch.cuda.synchronize()

    # ---- parity gate on this rank's batch (the timed batch): GPU vs oracle on the CPU-baseline sample
    cpu = None
    sample = min(args.cpu_sample_lines, n)
    if rank == 0 and not args.no_cpu_baseline:
        from oracle.oracle import OracleRegex  # the checker / reported baseline, never the measured path
        orx = OracleRegex(pattern)
        t0 = time.perf_counter()
        exp_caps, exp_status = orx.fullmatch_batch(data, off[:sample], length[:sample])
        cpu_s = time.perf_counter() - t0
        got_caps = d_caps[:sample].cpu().numpy()
        got_status = d_status[:sample].cpu().numpy()
        if not (np.array_equal(got_status, exp_status) and np.array_equal(got_caps, exp_caps)):
            raise SystemExit("PARITY FAILURE: GPU capture offsets differ from the oracle")
        # reported baseline: the reference processor's whole per-event work (regex_match + one SetContentNoCopy per
        # key + source tombstone + counters, oracle/processor_oracle.c) on the same lines, 1 thread = the reference's
        # default process_thread_count (core/app_config/AppConfig.cpp:58)
        keys = corpus.KEYS_A if args.regex == "A" else corpus.KEYS_B
        t0 = time.perf_counter()
        cnt = orx.process_batch(data, off[:sample], length[:sample], keys)
        cpu_proc_s = time.perf_counter() - t0
        assert cnt["out_successful"] == int(exp_status.sum())
        sample_bytes = float(length[:sample].sum())
        # all host cores, one slab of lines per thread (ctypes releases the GIL), as mReg[threadNo] would be used
        import concurrent.futures
        ncores = os.cpu_count() or 1
        slabs = [(i * sample // ncores, (i + 1) * sample // ncores) for i in range(ncores)]
        regs = [OracleRegex(pattern) for _ in range(ncores)]
        t0 = time.perf_counter()
        with concurrent.futures.ThreadPoolExecutor(ncores) as ex:
            list(ex.map(lambda a: regs[a[0]].process_batch(data, off[a[1][0]:a[1][1]], length[a[1][0]:a[1][1]], keys),
                        enumerate(slabs)))
        cpu_all_s = time.perf_counter() - t0
        cpu = {"value": round(sample_bytes / cpu_proc_s / 1e6, 1), "unit": "MB/s", "cores": 1, "kind": "port",
               "sample": "%d lines (%d MB) of the timed batch: oracle/bt_regex.c (boost::regex_match restated) + "
                         "oracle/processor_oracle.c (ProcessEvent work), 1 thread" % (sample, int(sample_bytes) >> 20),
               "match_only_MBps": round(sample_bytes / cpu_s / 1e6, 1),
               "all_cores": {"value": round(sample_bytes / cpu_all_s / 1e6, 1), "unit": "MB/s", "cores": ncores}}

    # ---- timed region: exactly K steps between barrier+synchronize pairs.  One HIP event pair on the launch stream
    # brackets the K back-to-back launches (per-launch event pairs insert markers between the kernels and were
    # measured to stretch the whole region); avg launch duration = event time / K.
    ev_start = torch.cuda.Event(enable_timing=True)
    ev_end = torch.cuda.Event(enable_timing=True)
    for _ in range(args.warmup):  # W untimed warm-up steps, immediately before the timed region (the CPU baseline
        step()                    # above leaves the GPU idle long enough for its clocks to drop)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev_start.record(stream)
    for _ in range(args.steps):
        step()
    ev_end.record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = [ev_start.elapsed_time(ev_end) / args.steps]

    matched = int(d_status.sum().item())
    from loongcollector_amd.shard import reduce_job
    # the job's only collective: MAX(elapsed) and SUM(counters) over ranks (RCCL); the data path has none
    elapsed, totals = reduce_job(elapsed, {"bytes": parsed_bytes_per_step * args.steps, "lines": n * args.steps,
                                           "matched_last": matched}, device=dev)

    if rank == 0:
        total_bytes = totals["bytes"]
        value = total_bytes / elapsed / 1e6
        avg_kernel_s = float(np.mean(kernel_ms)) / 1e3
        # algorithmic HBM bytes per line: payload L + 4 B offset + 1 B status + 8 B per group
        algo_bytes = (args.line_bytes + 5 + 8 * G) * n
        achieved = algo_bytes / avg_kernel_s / 1e9
        # HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of
        # this same command, profiles/round1_traffic.json); only quoted for the workload they were collected on
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "round1_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            if (tj.get("lines"), tj.get("regex"), tj.get("line_bytes"), tj.get("engine")) == (
                    n, args.regex, args.line_bytes, {1: "tdfa", 2: "nfa"}[info["engine"]]):
                traffic = tj["hbm_bytes_per_launch"]
        out = {
            "metric": "MB/s parsed (512B lines, 10-field regex) per MI355X + HBM-roofline %",
            "value": round(value, 1),
            "unit": "MB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "configs[1]: Apache-combined %dB lines, %d-group regex %s, %d-line batches resident in HBM"
                                   % (args.line_bytes, G, args.regex, n),
                       "engine": {1: "tdfa", 2: "nfa"}[info["engine"]], "lines_per_batch": n,
                       "tdfa_states": info["states"], "byte_classes": info["classes"],
                       "lds_table_bytes": info["table_bytes"], "parallelism": "line-shard x%d" % world,
                       "matched_lines_last_batch": matched},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                         "kernel": ("tdfa_match_kernel" if os.environ.get("LC_TDFA_STREAM") == "0" else "tdfa_stream_kernel") if info["engine"] == 1 else "nfa_match_kernel",
                         "avg_kernel_ms": round(avg_kernel_s * 1e3, 4),
                         "algorithmic_bytes_per_launch": algo_bytes},
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
