#!/usr/bin/env python3
"""bench.py -- headline benchmark: MB/s of log bytes parsed per MI355X (512 B Apache-combined lines, 10-field
regex, bit-exact capture offsets), with the HBM-roofline fraction and the host-CPU baseline next to it.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Default (BASELINE configs[1]).  One "step" = one pass of the hot path (lc_regex_match_device: the replacement for the
per-event BoostRegexMatch loop, core/plugin/processor/ProcessorParseRegexNative.cpp:115-124,194) over one batch of
1 Mi synthetic lines already resident in HBM.  `value` is that kernel-side rate; the same line also carries
`end_to_end` (the north-star path with the host in it: pinned staging + H2D || kernel || D2H through
lc_regex_match_host, and the in-agent shape -- lc_processor_process on 1000-line event groups from 1 and N runner
threads), measured in the same run, never inside the timed region.  Multi-GPU is an embarrassingly parallel line shard:
each rank owns its own batch, there is no data-path collective; the per-GPU counters are all-gathered (RCCL) so that
rank 0 can print the per-GPU table next to the aggregate.  Rank 0 prints ONE JSON line.

    python bench.py --config 4     # BASELINE configs[3]: 64 pipelines, each its own regex, ~512 KB groups round-robin
    python bench.py --config 5     # BASELINE configs[4]: mixed nginx/JSON corpus fed FROM HOST in 64 MiB slabs per rank
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
PCIE_GEN5_X16_GBPS = 63.0  # PCIe 5.0 x16, one direction, after 128b/130b encoding


PLACEMENT = {}


def compact_tables(rx):
    from loongcollector_amd import binding
    try:
        blob = rx.table(binding.LC_TABLE_TDFA_WIDE_BLOB, np.uint32)
        if blob is None or len(blob) < 16:
            return None
        po = int(blob[7]) // 4
        return {"block": int(blob[15]), "bytes": int(blob.nbytes),
                "pair_table": None if not po else ("one stamp per pair" if int(blob[po + 4]) == 1 else "two stamps per pair")}
    except Exception:  # noqa: BLE001 -- informational
        return None


def setup_dist():
    import torch
    import torch.distributed as dist
    from loongcollector_amd.shard import place_rank
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the parse engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # this process owns ONE GPU: every thread that enters the library's host entry points (the runner threads of the in-agent legs,
    # the slab feeders) goes to it -- the library's own default would deal threads over every visible device
    from loongcollector_amd import binding as _binding
    _binding.set_bind_policy(_binding.LC_BIND_FIXED, local_rank)
    # with several ranks on one host every rank stays on the CPUs / memory of its GPU's NUMA node: set BEFORE any pinned
    # staging memory is allocated (first touch).  One rank keeps the whole box (its runner-thread measurements want the cores).
    PLACEMENT.update(place_rank(local_rank, apply=world > 1))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)
    return world, rank, dev


# ------------------------------------------------------------------------------------------------- end to end (host in the path)
def measure_h2d(dev, nbytes=256 << 20, reps=5):
    """Raw pinned host -> device copy rate on this box (GB/s): what bounds every host-fed path."""
    import torch
    h = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        d.copy_(h, non_blocking=True)
    b.record()
    torch.cuda.synchronize()
    return nbytes * reps / (a.elapsed_time(b) * 1e-3) / 1e9


def measure_in_agent(pattern, keys, data, off, length, group_lines, n_groups, threads):
    """lc_processor_process (gather views -> pinned staging -> H2D -> kernel -> D2H -> zero-copy stitch + policy) on event
    groups of `group_lines` lines, the shape ProcessorRunner hands over (core/runner/ProcessorRunner.cpp:138-142), driven by
    `threads` runner threads that share ONE processor instance (as the agent's threads share one plugin instance).
    -> MB/s of `content` bytes.  ctypes releases the GIL for the duration of the C call."""
    import concurrent.futures
    from loongcollector_amd.processor import EventGroup, Processor
    cfg = {"SourceKey": "content", "Regex": pattern, "Keys": keys}
    proc = Processor(cfg)
    groups, nbytes = [], 0
    for g in range(n_groups):
        lo = g * group_lines
        groups.append(EventGroup.from_lines(data, off[lo:lo + group_lines], length[lo:lo + group_lines]))
        nbytes += int(length[lo:lo + group_lines].sum())
    import threading
    slices = [groups[t::threads] for t in range(threads)]
    warm = [EventGroup.from_lines(data, off[:group_lines], length[:group_lines]) for _ in range(threads)]
    gate = threading.Barrier(threads + 1)
    ends = [0.0] * threads

    def run(t):
        proc.process(warm[t])  # this thread's first call allocates its pinned staging and stream: not part of the figure
        gate.wait()
        for g in slices[t]:
            proc.process(g)
        ends[t] = time.perf_counter()

    with concurrent.futures.ThreadPoolExecutor(threads) as ex:
        futs = [ex.submit(run, t) for t in range(threads)]
        gate.wait()
        t0 = time.perf_counter()
        for f in futs:
            f.result()
    dt = max(ends) - t0
    c = proc.counters()
    lines = n_groups * group_lines + group_lines * threads
    if c["out_successful_events_total"] != lines or c["out_failed_events_total"] != 0:
        raise SystemExit("PARITY FAILURE (in-agent path): %d of %d events parsed" % (c["out_successful_events_total"], lines))
    first = groups[0].contents()[0]
    return nbytes / dt / 1e6, first


def measure_in_agent_window(pattern, keys, data, off, length, group_lines, n_groups, threads, window=16, _types=None):
    """The in-agent shape with a BOUNDED number of groups alive, which is what an agent's queues allow: per round, `window` groups per
    runner thread are built (the reader's job, untimed), processed by the `threads` runner threads sharing one instance (timed: from
    the moment the round opens to the last thread's end), dropped (the flusher's job, untimed).  measure_in_agent() keeps every group
    of the run alive, so each arena chunk a stitch takes is memory nobody has touched yet (108 first-touch page faults per 1000-event
    group); here the chunks of the groups dropped a round ago come back through the event model's pool (csrc/event_model.hpp
    ArenaChunkPool).  The first round fills the pool and is not timed.  -> (MB/s of `content` bytes, fields of the first event)."""
    import concurrent.futures
    import threading
    if _types is None:
        from loongcollector_amd.processor import EventGroup, Processor
    else:
        EventGroup, Processor = _types   # (tests/test_bench_launch.py drives the round logic without a device)
    proc = Processor({"SourceKey": "content", "Regex": pattern, "Keys": keys})
    per_round = window * threads
    rounds = max(2, n_groups // per_round) + 1
    gate = threading.Barrier(threads + 1)
    state = {"groups": [], "failed": None}
    starts = [[0.0] * threads for _ in range(rounds)]
    ends = [[0.0] * threads for _ in range(rounds)]

    def build(r):
        out = []
        for k in range(per_round):
            lo = ((r * per_round + k) % n_groups) * group_lines
            out.append(EventGroup.from_lines(data, off[lo:lo + group_lines], length[lo:lo + group_lines]))
        return out

    def run(t):
        try:
            w = EventGroup.from_lines(data, off[:group_lines], length[:group_lines])
            proc.process(w)   # this thread's first call allocates its pinned staging and stream: not part of the figure
            w.close()
            for r in range(rounds):
                gate.wait(timeout=300)
                starts[r][t] = time.perf_counter()
                for g in state["groups"][t * window:(t + 1) * window]:
                    proc.process(g)
                ends[r][t] = time.perf_counter()
                gate.wait(timeout=300)
        except BaseException as e:   # (a broken barrier must not leave the other side waiting for ever)
            state["failed"] = state["failed"] or e
            gate.abort()
            raise

    timed, nbytes, first, lines = 0.0, 0, None, group_lines * threads
    with concurrent.futures.ThreadPoolExecutor(threads) as ex:
        futs = [ex.submit(run, t) for t in range(threads)]
        try:
            for r in range(rounds):
                state["groups"] = build(r)
                gate.wait(timeout=300)   # the round opens
                gate.wait(timeout=300)   # ... and is over
                lines += per_round * group_lines
                if r == 0:
                    first = state["groups"][0].contents()[0]
                else:
                    timed += max(ends[r]) - min(starts[r])
                    for k in range(per_round):
                        lo = ((r * per_round + k) % n_groups) * group_lines
                        nbytes += int(length[lo:lo + group_lines].sum())
                for g in state["groups"]:
                    g.close()
        except threading.BrokenBarrierError:
            pass
        concurrent.futures.wait(futs)
    if state["failed"] is not None:   # the first thing that went wrong, not the broken barriers it left behind
        raise state["failed"]
    for f in futs:
        f.result()
    c = proc.counters()
    if c["out_successful_events_total"] != lines or c["out_failed_events_total"] != 0:
        raise SystemExit("PARITY FAILURE (in-agent path, bounded window): %d of %d events parsed" % (c["out_successful_events_total"], lines))
    return nbytes / timed / 1e6, first


def measure_in_agent_reference_shape(thread_counts, dev, lines, group_lines):
    """the in-agent leg on the reference-shaped build of the library, in a process of its own (tools/inagent_shape_bench.py)"""
    import subprocess
    from loongcollector_amd import build as native_build
    if not os.path.exists(native_build.LIB_REFSHAPE):
        return {"error": "liblc_regex_gpu_refshape.so has not been built"}
    env = dict(os.environ, LC_REGEX_GPU_LIB=native_build.LIB_REFSHAPE)
    env["LC_BIND_POLICY"] = "fixed:%d" % (dev.index or 0)
    cmd = [sys.executable, os.path.join(ROOT, "tools", "inagent_shape_bench.py"), "--threads", ",".join(str(t) for t in thread_counts),
           "--lines", str(lines), "--group-lines", str(group_lines), "--device", str(dev.index or 0)]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    except subprocess.TimeoutExpired:
        return {"error": "timeout"}
    if r.returncode != 0:
        if "PARITY FAILURE" in r.stderr + r.stdout:
            raise SystemExit((r.stderr + r.stdout).strip().splitlines()[-1])
        return {"error": (r.stderr or r.stdout).strip()[-300:]}
    d = json.loads(r.stdout.strip().splitlines()[-1])
    return {"window16": d["window16_MBps"], "all_groups_alive": d["all_groups_alive_MBps"], "lib": d["lib"],
            "window16_one_thread": {"median_of": d.get("window16_one_thread_runs_MBps", [])}}


def measure_pipeline(thread_counts, buffer_bytes=512 << 10, n_buffers=64):
    """The reference's benchmark pipeline (test/benchmark/local/test_cases/performance_file_to_blackhole_loongcollector/
    loongcollector.yaml: split -> processor_parse_regex_native with regex B -> processor_filter_regex_native on user_agent) on
    512 KB read buffers, the unit LogFileReader hands over: lc_pipeline_process = ONE device trip per buffer, only the survivors
    come back and become events.  -> MB/s of raw buffer bytes per thread count, next to the same three steps run one after the
    other (three gathers, three trips) and the reference's published 68 MB/s."""
    import concurrent.futures
    import threading
    from loongcollector_amd import corpus
    from loongcollector_amd.processor import EventGroup, Pipeline
    parse = {"SourceKey": "content", "Regex": corpus.REGEX_B, "Keys": corpus.KEYS_B}
    filt = {"FilterKey": ["user_agent"], "FilterRegex": ["^no-agent$"]}
    data, off, length = corpus.apache_batch(buffer_bytes // 513 * 4, "B", 512, seed=corpus.SEED + 99, pool_lines=2048)
    lines = [bytes(data[o:o + l]) for o, l in zip(off[:-1], length)]
    for i in range(0, len(lines), 50):   # 2 % of the lines carry the agent the filter keeps
        head, _, _ = lines[i].rpartition(b' "')
        lines[i] = head + b' "no-agent"'
    per = buffer_bytes // 513
    buffers = [b"\n".join(lines[k * per:(k + 1) * per]) + b"\n" for k in range(4)]
    out = {"what": "lc_pipeline_process on %d KB read buffers (%d lines of 512 B): split -> parse (regex B, 11 keys) -> filter "
                   "user_agent ^no-agent$ (2 %% of the lines survive), one device trip per buffer, N runner threads sharing one "
                   "instance" % (buffer_bytes >> 10, per),
           "reference_MBps": 68.0, "reference_what": "LoongCollector's published single-thread figure for this pipeline (BASELINE.md)"}
    expect = None
    # (round 6: the three-steps-in-a-row form -- "Fused": false, what a pipeline that cannot travel fused falls back to -- is no longer a
    # leg of this line: it was an A/B of the fused trip, and its 16 -> 32 thread figures said nothing the fused leg does not)
    for label, fused in (("fused_MBps", True),):
        pipe = Pipeline({"Parse": parse, "Filter": filt, "Fused": fused})
        assert pipe.fused == fused
        res = {}
        for t in thread_counts:
            per_thread = max(8, n_buffers // t)
            gate = threading.Barrier(t + 1)
            ends = [0.0] * t
            kept = [0] * t

            def run(tid):
                mine = [EventGroup.from_buffer(buffers[(tid + k) % len(buffers)], file_offset=k * buffer_bytes) for k in range(per_thread + 1)]
                pipe.process(mine[0])  # this thread's first call allocates its staging and stream
                gate.wait()
                for g in mine[1:]:
                    pipe.process(g)
                    kept[tid] += len(g)
                ends[tid] = time.perf_counter()

            with concurrent.futures.ThreadPoolExecutor(t) as ex:
                futs = [ex.submit(run, tid) for tid in range(t)]
                gate.wait()
                t0 = time.perf_counter()
                for f in futs:
                    f.result()
            dt = max(ends) - t0
            nbytes = sum(len(buffers[(tid + k) % len(buffers)]) for tid in range(t) for k in range(1, per_thread + 1))
            res[str(t)] = round(nbytes / dt / 1e6, 1)
            survivors = sum(kept)
            want = sum(sum(1 for ln in buffers[(tid + k) % len(buffers)].split(b"\n") if ln.endswith(b'"no-agent"'))
                       for tid in range(t) for k in range(1, per_thread + 1))
            if survivors != want:
                raise SystemExit("PARITY FAILURE (pipeline, %s): %d events survived, %d lines carry the kept agent" % (label, survivors, want))
        out[label] = res
        c = pipe.counters()
        if expect is None:
            expect = (c["out_failed_events_total"], c["discarded_events_total"])
        out.setdefault("groups", {})[label] = {"fused": c["groups_fused"], "chained": c["groups_chained"]}
    return out


def _run_threads(threads, per_thread, make_work, do_work):
    """t runner threads, each with its own work items (the first is its warm-up: staging and stream are allocated there)
    -> seconds for per_thread items per thread"""
    import concurrent.futures
    import threading
    gate = threading.Barrier(threads + 1)
    ends = [0.0] * threads

    def run(tid):
        mine = [make_work(tid, k) for k in range(per_thread + 1)]
        do_work(tid, mine[0])
        gate.wait()
        for w in mine[1:]:
            do_work(tid, w)
        ends[tid] = time.perf_counter()

    with concurrent.futures.ThreadPoolExecutor(threads) as ex:
        futs = [ex.submit(run, tid) for tid in range(threads)]
        gate.wait()
        t0 = time.perf_counter()
        for f in futs:
            f.result()
    return max(ends) - t0


def measure_multiline(thread_counts, buffer_bytes=512 << 10, n_buffers=48):
    """The multiline splitter (ProcessorSplitMultilineLogStringNative) on 512 KB read buffers of Java stack traces:
    lc_multiline_split_host = ONE device trip per buffer (upload, split kernels, one status-only launch per pattern, the
    start/continue/end walk as a scan on the device), only the records come back.  Next to the reference's published 238 MB/s
    for multi-line collection (README.md:66-67).  Parity gate: the records of every distinct buffer equal the oracle's."""
    from loongcollector_amd import corpus
    from loongcollector_amd.multiline import Multiline
    from oracle.multiline_oracle import MultilineOracle
    cfg = {"StartPattern": corpus.MULTILINE_START}
    buffers = [corpus.multiline_buffer(buffer_bytes, seed=corpus.SEED + 17 + k, unmatched_head=3 * k) for k in range(4)]
    m = Multiline(**cfg)
    o = MultilineOracle(**cfg)
    logs = 0
    for b in buffers:
        got = m.split(b)
        if got != o.split(b):
            raise SystemExit("PARITY FAILURE (multiline): records differ from the oracle's walk")
        logs += got[1][2]
    out = {"what": "lc_multiline_split_host on %d KB read buffers of stack traces (StartPattern %s; %d lines, %d logs per buffer): "
                   "upload -> split kernels -> status-only match -> record scan on the device, records come back; N runner threads "
                   "sharing one instance" % (buffer_bytes >> 10, corpus.MULTILINE_START, buffers[0].count(b"\n"), logs // len(buffers)),
           "reference_MBps": 238.0, "reference_what": "LoongCollector's published multi-line figure (README.md:66-67), whole agent, 1 thread"}
    res = {}
    for t in thread_counts:
        per_thread = max(8, n_buffers // t)
        dt = _run_threads(t, per_thread, lambda tid, k: buffers[(tid + k) % len(buffers)], lambda tid, b: m.split_count(b))
        nbytes = sum(len(buffers[(tid + k) % len(buffers)]) for tid in range(t) for k in range(1, per_thread + 1))
        res[str(t)] = round(nbytes / dt / 1e6, 1)
    out["MBps"] = res
    # three patterns (start + continue + end all in use): three status launches over the same device copy
    cfg3 = {"StartPattern": corpus.MULTILINE_START, "ContinuePattern": r"(\tat |Caused by: ).*", "EndPattern": r"\tat \S+\(Handler\.java:\d+\)$"}
    m3, o3 = Multiline(**cfg3), MultilineOracle(**cfg3)
    if m3.split(buffers[1]) != o3.split(buffers[1]):
        raise SystemExit("PARITY FAILURE (multiline, three patterns): records differ from the oracle's walk")
    dt = _run_threads(1, 16, lambda tid, k: buffers[k % len(buffers)], lambda tid, b: m3.split_count(b))
    out["three_patterns_MBps"] = {"1": round(sum(len(buffers[k % len(buffers)]) for k in range(1, 17)) / dt / 1e6, 1)}
    return out


def measure_filter(thread_counts, group_lines=1000, n_groups=128):
    """processor_filter_regex_native on event groups of 1000 parsed events (the keys of regex B's captures): a ConditionExp of three
    regex leaves -- ONE device trip per group (one upload of the three keys' values, the leaves as jobs of one packed launch, one
    copy back).  Parity gate: the surviving events equal oracle/filter_oracle.py's."""
    from loongcollector_amd import corpus
    from loongcollector_amd.processor import EventGroup, Filter
    from oracle.filter_oracle import FilterOracle
    cond = {"operator": "and", "operands": [
        {"operator": "or", "operands": [{"key": "method", "exp": "GET|HEAD", "type": "regex"},
                                        {"key": "response_code", "exp": "5\\d\\d", "type": "regex"}]},
        {"operator": "not", "operands": [{"key": "user_agent", "exp": ".*(bot|spider|crawl).*", "type": "regex"}]}]}
    config = {"ConditionExp": cond}
    data, off, length = corpus.apache_batch(group_lines * 4, "B", 512, seed=corpus.SEED + 5, pool_lines=2048)
    from oracle.oracle import OracleRegex
    rx = OracleRegex(corpus.REGEX_B)
    caps, status = rx.fullmatch_batch(data, off[:-1], length)
    events = []
    for i in range(len(length)):
        raw = data[int(off[i]):int(off[i]) + int(length[i])].tobytes()
        events.append({"contents": [[k, raw[caps[i][2 * g]:caps[i][2 * g + 1]].decode("latin-1")] for g, k in enumerate(corpus.KEYS_B)],
                       "timestamp": 1, "type": 1})
    fixtures = [{"events": events[k * group_lines:(k + 1) * group_lines]} for k in range(4)]
    value_bytes = [sum(len(v) for ev in fx["events"] for k, v in ev["contents"] if k in ("method", "response_code", "user_agent")) for fx in fixtures]
    group_bytes = [sum(len(k) + len(v) for ev in fx["events"] for k, v in ev["contents"]) for fx in fixtures]
    f = Filter(config)
    fo = FilterOracle(config)
    for fx in fixtures:
        g = EventGroup(fx)
        f.process(g)
        got = [dict(ev) for ev in g.contents()]
        want = [{k: v.decode("latin-1") for k, v in c.items()}
                for c in fo.process([{k: v.encode("latin-1") for k, v in ev["contents"]} for ev in fx["events"]])]
        if got != want or not 0 < len(want) < len(fx["events"]):
            raise SystemExit("PARITY FAILURE (filter): surviving events differ from the oracle's")
    out = {"what": "lc_filter_process on groups of %d parsed events (11 keys); ConditionExp (method ~ GET|HEAD or response_code ~ 5xx) and "
                   "not user_agent ~ bot: three regex leaves = ONE device trip per group; N runner threads sharing one instance; "
                   "MB/s of the groups' content bytes (the three keys' values that go up are %.0f %% of them)"
                   % (group_lines, 100.0 * sum(value_bytes) / sum(group_bytes))}
    res = {}
    for t in thread_counts:
        # (at least 16 timed groups per runner thread: with 4 -- 128 groups over 32 threads -- the timed window was a millisecond, of the
        # order of the time Python needs to release 33 threads from the barrier, and the 32-thread figure measured that)
        per_thread = max(16, n_groups // t)
        dt = _run_threads(t, per_thread, lambda tid, k: EventGroup(fixtures[(tid + k) % 4]), lambda tid, g: f.process(g))
        nbytes = sum(group_bytes[(tid + k) % 4] for tid in range(t) for k in range(1, per_thread + 1))
        res[str(t)] = round(nbytes / dt / 1e6, 1)
    out["MBps"] = res
    return out


def end_to_end(rx, pattern, keys, data, off, length, exp_caps, dev, thread_counts, group_lines=1000, e2e_lines=1 << 18):
    G = rx.groups
    n = len(length)
    out = {}
    h2d = measure_h2d(dev)
    out["h2d_GBps"] = round(h2d, 1)
    # -- the host entry point: contiguous lines, two pinned slots, two streams (copy of chunk k+1 || kernel of chunk k)
    rx.match_host(data, off[:-1], length)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        caps, status = rx.match_host(data, off[:-1], length)
        best = min(best, time.perf_counter() - t0)
    if exp_caps is not None and not np.array_equal(caps[:len(exp_caps)], exp_caps):
        raise SystemExit("PARITY FAILURE (host path): capture offsets differ from the oracle")
    payload = float(length.sum())
    out["host_MBps"] = round(payload / best / 1e6, 1)
    up = float(off[n]) + 8.0 * n  # bytes that cross the bus upwards per call: lines + separators, (offset, length) per line
    out["host_h2d_GBps"] = round(up / best / 1e9, 1)
    out["pcie_frac"] = round(up / best / 1e9 / h2d, 3)
    out["host_what"] = "lc_regex_match_host: %d lines gathered into 2 pinned slots, H2D || kernel || D2H on 2 streams, 1 host thread" % n
    # -- the in-agent shape
    m = min(e2e_lines, n) // group_lines * group_lines
    ag, ag_all_alive = {}, {}
    for t in thread_counts:
        mbps, first = measure_in_agent_window(pattern, keys, data, off, length, group_lines, m // group_lines, t)
        ag[str(t)] = round(mbps, 1)
    # (round 3's form of the leg -- every group of the run alive, each stitch on untouched memory -- beside it, one thread)
    for t in thread_counts:
        ag_all_alive[str(t)] = round(measure_in_agent(pattern, keys, data, off, length, group_lines, m // group_lines, t)[0], 1)
    if exp_caps is not None:  # spot check of the stitch: the first event's fields are the oracle's captures of line 0
        raw = data[int(off[0]):int(off[0]) + int(length[0])].tobytes()
        want = [(k, raw[exp_caps[0][2 * i]:exp_caps[0][2 * i + 1]].decode("latin-1")) for i, k in enumerate(keys)]
        if [tuple(kv) for kv in first] != want:
            raise SystemExit("PARITY FAILURE (in-agent path): stitched fields differ from the oracle's captures")
    # the bounded-window figure has its own key; `in_agent_MBps` stays what it was through round 3 (every group of the run
    # alive) so that rounds compare like for like.  Both are measured on the library's OWN event model (csrc/event_model.hpp: arena
    # vector, chunk pool, one-call stitch) -- an agent build has the reference's LogEvent instead: in_agent_reference_shape_MBps.
    out["in_agent_window16_MBps"] = ag
    out["in_agent_all_groups_alive_MBps"] = ag_all_alive
    out["in_agent_MBps"] = dict(ag_all_alive)
    out["in_agent_reference_shape_MBps"] = measure_in_agent_reference_shape(thread_counts, dev, m, group_lines)
    # -- the same groups through the columnar entry (lc_processor_parse_columnar: capture table + base pointers + per-event protobuf
    # content sizes, no event materialised): what a serializer downstream can consume directly
    from loongcollector_amd.processor import EventGroup, Processor
    colp = Processor({"SourceKey": "content", "Regex": pattern, "Keys": keys})
    col = {}
    for t in thread_counts:
        per_thread = max(8, (m // group_lines) // t)

        def mk(tid, k):
            lo = ((tid * per_thread + k) * group_lines) % (m - group_lines + 1)
            return EventGroup.from_lines(data, off[lo:lo + group_lines], length[lo:lo + group_lines])

        seen = []
        counted = set()

        def work(tid, g):
            # (the first group of every thread is counted -- events, parsed events -- through numpy views of the table; the others are
            # parse + free only: the conversions cost 60-80 us per group under the GIL, and from 16 threads on they, not the library,
            # were what this leg measured: round 4's "columnar dips past 16 threads")
            if tid not in counted:
                counted.add(tid)
                seen.append(colp.parse_columnar_count(g))
            else:
                colp.parse_columnar_discard(g)

        dt = _run_threads(t, per_thread, mk, work)
        if any(n != group_lines or ok != group_lines for n, ok, _ in seen):
            raise SystemExit("PARITY FAILURE (columnar path): not every event of a group was parsed")
        col[str(t)] = round(t * per_thread * group_lines * float(length[:group_lines].mean()) / dt / 1e6, 1)
    out["in_agent_columnar_MBps"] = col
    out["pipeline"] = measure_pipeline(thread_counts)
    out["multiline"] = measure_multiline(thread_counts)
    out["filter"] = measure_filter(thread_counts)
    out["in_agent_what"] = ("lc_processor_process on %d-line event groups (gather into pinned staging -> ONE kernel launch that reads the lines and writes the capture table through the pinned mapping -> stitch + policy), "
                            "N runner threads sharing one instance; %d lines.  in_agent_window16_MBps: 16 groups alive per thread (built before and dropped after each timed round: the reader's and the "
                            "flusher's job), first round untimed; in_agent_MBps = in_agent_all_groups_alive_MBps: every group of the run alive (rounds 1-3's method).  BOTH run on the library's stand-in "
                            "event model (csrc/event_model.hpp: arena-backed contents, chunk pool, one-call stitch AppendCapturesNoCopy), which an agent build does NOT have.  "
                            "in_agent_reference_shape_MBps: the same leg on liblc_regex_gpu_refshape.so -- the event model in the reference's shape (heap std::vector contents, no pool, "
                            "K x SetContentNoCopy + DelContent as LogEvent.cpp:83-106 offers them) -- i.e. what the reference's LogEvent lets a runner thread do; tools/stitch_bench.py compares "
                            "that shape with the reference's own LogEvent.cpp compiled from source (within 10 %% on the build machine)" % (group_lines, m))
    return out


# ------------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(pattern, keys, data, off, length, sample, d_caps, d_status):
    from oracle import oracle as O  # the checker / reported baseline, never the measured path
    orx = O.OracleRegex(pattern)
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
    t0 = time.perf_counter()
    cnt = orx.process_batch(data, off[:sample], length[:sample], keys)
    cpu_proc_s = time.perf_counter() - t0
    assert cnt["out_successful"] == int(exp_status.sum())
    sample_bytes = float(length[:sample].sum())
    # all host cores, one slab of lines per thread (ctypes releases the GIL), as mReg[threadNo] would be used
    import concurrent.futures
    ncores = os.cpu_count() or 1
    slabs = [(i * sample // ncores, (i + 1) * sample // ncores) for i in range(ncores)]
    regs = [O.OracleRegex(pattern) for _ in range(ncores)]
    t0 = time.perf_counter()
    with concurrent.futures.ThreadPoolExecutor(ncores) as ex:
        list(ex.map(lambda a: regs[a[0]].process_batch(data, off[a[1][0]:a[1][1]], length[a[1][0]:a[1][1]], keys),
                    enumerate(slabs)))
    cpu_all_s = time.perf_counter() - t0
    # the stand-in engine BASELINE.md section 2 names: PCRE1 8.45 behind the same full-match wrapper (match only), checked
    # against the oracle's captures on the lines it is timed on
    engines = [{"engine": "oracle/bt_regex.c", "match_only_MBps": round(sample_bytes / cpu_s / 1e6, 1), "cores": 1}]
    ps = min(sample, 1 << 18)
    pv = O.pcre_version()
    for jit in (False, True):
        if not pv:
            break
        t0 = time.perf_counter()
        r = O.pcre_fullmatch_batch(pattern, data, off[:ps], length[:ps], orx.groups, jit)
        dt = time.perf_counter() - t0
        engines.append({"engine": "PCRE1 %s%s" % (pv.split()[0], " JIT" if jit else ""), "cores": 1,
                        "match_only_MBps": round(float(length[:ps].sum()) / dt / 1e6, 1), "lines": ps,
                        "captures_equal_oracle": bool(np.array_equal(r[0], exp_caps[:ps]) and np.array_equal(r[1], exp_status[:ps]))})
    cpu = {"value": round(sample_bytes / cpu_proc_s / 1e6, 1), "unit": "MB/s", "cores": 1, "kind": "port",
           "engine": "oracle/bt_regex.c: backtracking matcher restating boost::regex_match, with counted fast paths (faster than "
                     "PCRE1, see engines; boost 1.68 itself is not available on this box)",
           "sample": "%d lines (%d MB) of the timed batch: oracle/bt_regex.c (boost::regex_match restated) + "
                     "oracle/processor_oracle.c (ProcessEvent work), 1 thread" % (sample, int(sample_bytes) >> 20),
           "match_only_MBps": round(sample_bytes / cpu_s / 1e6, 1),
           "all_cores": {"value": round(sample_bytes / cpu_all_s / 1e6, 1), "unit": "MB/s", "cores": ncores},
           "engines": engines}
    return cpu, exp_caps


# ------------------------------------------------------------------------------------------------- configs[1]: the headline
def run_headline(args):
    import ctypes

    import torch
    import torch.distributed as dist

    from loongcollector_amd import binding, corpus
    from loongcollector_amd.shard import gather_job

    world, rank, dev = setup_dist()
    pattern = corpus.REGEX_A if args.regex == "A" else corpus.REGEX_B
    keys = corpus.KEYS_A if args.regex == "A" else corpus.KEYS_B
    engine = {"auto": binding.LC_ENGINE_AUTO, "tdfa": binding.LC_ENGINE_TDFA, "nfa": binding.LC_ENGINE_NFA}[args.engine]
    rx = binding.GpuRegex(pattern, engine=engine)
    G = rx.groups
    info = rx.info()

    n = args.lines
    # SURVEY.md section 8(d)'s recipe: every line generated on its own from one std::mt19937_64 stream, seed 20260921 (+ 1000 per rank);
    # no line occurs twice (through round 4: 1 Mi draws from a pool of 8 192 numpy-made lines)
    data, off, length = corpus.apache_lines(n, args.regex, args.line_bytes, seed=corpus.SEED + 1000 * rank)
    parsed_bytes_per_step = int(length.sum())
    d_data = torch.from_numpy(data).to(dev)
    d_off = torch.from_numpy(off.view(np.int32)).to(dev)
    d_caps = torch.empty((n, 2 * G), dtype=torch.int32, device=dev)
    d_status = torch.empty((n,), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream()

    # one step == one lc_regex_match_device_engine() call; arguments are marshalled once so that the timed loop is
    # the C-ABI call itself and not Python argument conversion
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

    binding.launched_kernels()
    step()  # one untimed pass to produce the capture table the parity gate below checks
    torch.cuda.synchronize()
    kernels = binding.launched_kernels()

    # ---- parity gate on this rank's batch (the timed batch): GPU vs oracle on the CPU-baseline sample
    cpu, exp_caps = None, None
    sample = min(args.cpu_sample_lines, n)
    if rank == 0 and not args.no_cpu_baseline:
        cpu, exp_caps = cpu_baseline(pattern, keys, data, off, length, sample, d_caps, d_status)

    # ---- the host-inclusive figures (rank 0 only, outside the timed region, bounded: a few seconds)
    e2e = None
    if rank == 0 and not args.no_e2e:
        ncores = os.cpu_count() or 1
        tc = [1] + ([min(16, ncores)] if ncores > 1 else []) + ([min(32, ncores)] if ncores > 16 else [])
        e2e = end_to_end(rx, pattern, keys, data, off, length, exp_caps, dev, tc)

    # ---- timed region: exactly K steps between barrier+synchronize pairs.  One HIP event pair on the launch stream
    # brackets the K back-to-back launches (per-launch event pairs insert markers between the kernels and were
    # measured to stretch the whole region); avg launch duration = event time / K.
    ev_start = torch.cuda.Event(enable_timing=True)
    ev_end = torch.cuda.Event(enable_timing=True)
    # The CPU baseline and the host-path measurements above leave the GPU idle for tens of seconds: its clocks drop, and W = 3
    # warm-up steps (0.8 ms) do not bring them back -- the same kernel then measures 7 % slower than under rocprofv3, which
    # runs it without that pause.  A short untimed spin restores the clocks; then the W warm-up steps, then the timed K.
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.25:
        for _ in range(32):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):  # W untimed warm-up steps, immediately before the timed region
        step()
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
    kernel_ms = ev_start.elapsed_time(ev_end) / args.steps

    matched = int(d_status.sum().item())
    # ---- the other BASELINE configs, bounded, AFTER the timed region (never inside it): so that the driver's one line is
    # evidence for all five configs.  configs[4] is host-fed and collective (barriers + the counters' all-gather): every rank runs it.
    extra = {}
    if not args.no_configs:
        import argparse as _ap
        sub = _ap.Namespace(**vars(args))
        # (4 GiB per rank = 64 slabs: a bounded run whose fill and drain are 10 % of it, not 40 %; ~0.1 s)
        sub.corpus_gb, sub.slab_mib, sub.no_cpu_baseline = 4.0 * world, 64, args.no_cpu_baseline
        host_fed = compute_sharded_corpus(sub, world, rank, dev)
        if rank == 0:
            extra["configs[4]"] = host_fed
        if rank == 0 and world == 1:
            sub = _ap.Namespace(**vars(args))
            sub.pipelines, sub.group_lines, sub.streams, sub.steps, sub.warmup = 64, 1000, 4, 20, 3
            extra["configs[3]"] = compute_multitenant(sub, dev)
            # configs[2] runs in a process of its own: a process that hosts a Grok processor asks the HIP runtime for 16 hardware
            # queues (csrc/gpu_runtime.hip lcPreferHwQueuesForGrok; the runtime reads GPU_MAX_HW_QUEUES when it initialises, and
            # torch has initialised it in THIS process long ago).  The other legs keep the runtime's default: 16 queues cost the
            # one-stream-per-runner-thread paths 15-20 % (profiles/round3_grok_streams.txt).
            import subprocess
            env = dict(os.environ)
            env.setdefault("GPU_MAX_HW_QUEUES", "16")
            env["LC_BIND_POLICY"] = "fixed:%d" % (dev.index or 0)  # (one GPU per process: see setup_dist)
            try:
                out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "grok_config2.py"), "--device", str(dev.index or 0)],
                                     env=env, capture_output=True, text=True, timeout=900)
                rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
                if out.returncode != 0 or not rows:  # (reported, not fatal: the headline line must not depend on this leg)
                    extra["configs[2]"] = {"error": "tools/grok_config2.py failed (rc %d): %s" % (out.returncode, out.stderr[-600:])}
                else:
                    extra["configs[2]"] = rows[-1]
            except subprocess.TimeoutExpired:
                extra["configs[2]"] = {"error": "tools/grok_config2.py did not finish in 900 s"}
            except Exception as ex:  # noqa: BLE001 -- reported, not fatal
                extra["configs[2]"] = {"error": "tools/grok_config2.py: %r" % (ex,)}
            # (round 6) patterns no automaton runs -- back-references, general look-arounds: the device backtracking engine beside the
            # tagged DFA on the headline's line shape (tools/bt_bench.py; no BASELINE config holds such a pattern, so this is a leg
            # beside the configs, bounded to 65 536 lines; every leg's first 4 096 lines are compared with the oracle before it is timed)
            try:
                out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bt_bench.py"), "--lines", "65536", "--steps", "5"],
                                     env=dict(os.environ), capture_output=True, text=True, timeout=300)
                rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
                extra["backtracking_engine"] = ({"what": "LC_ENGINE_BT (csrc/bt_vm.hpp): one line per lane, program in LDS, explicit stack in HBM; "
                                                          "512 B Apache lines resident in HBM", "legs": rows}
                                                 if out.returncode == 0 and rows else
                                                 {"error": "tools/bt_bench.py failed (rc %d): %s" % (out.returncode, (out.stdout + out.stderr)[-400:])})
            except Exception as ex:  # noqa: BLE001 -- reported, not fatal
                extra["backtracking_engine"] = {"error": "tools/bt_bench.py: %r" % (ex,)}
    # the job's only collective: ONE all-gather of the per-GPU counters (RCCL); the data path has none
    per_gpu = gather_job({"bytes": parsed_bytes_per_step * args.steps, "lines": n * args.steps, "matched_last": matched,
                          "elapsed_us": int(elapsed * 1e6), "kernel_us_per_step": int(kernel_ms * 1e3)}, device=dev)

    if rank == 0:
        elapsed = max(g["elapsed_us"] for g in per_gpu) / 1e6
        total_bytes = sum(g["bytes"] for g in per_gpu)
        value = total_bytes / elapsed / 1e6
        avg_kernel_s = kernel_ms / 1e3
        # algorithmic HBM bytes per line: payload L + 4 B offset + 1 B status + 8 B per group
        algo_bytes = (args.line_bytes + 5 + 8 * G) * n
        achieved = algo_bytes / avg_kernel_s / 1e9
        # HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of
        # this same command, profiles/round*_traffic.json); only quoted for the workload they were collected on
        traffic, traffic_source = None, None
        # (the newest round's file; quoted only when it was collected on this workload AND on the kernels this run launched)
        import glob
        for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_traffic.json")), reverse=True):
            with open(tpath) as f:
                tj = json.load(f)
            same_kernels = tj.get("kernels_launched") in (None, kernels)
            if same_kernels and (tj.get("lines"), tj.get("regex"), tj.get("line_bytes"), tj.get("engine")) == (
                    n, args.regex, args.line_bytes, {1: "tdfa", 2: "nfa"}[info["engine"]]):
                traffic = tj["hbm_bytes_per_launch"]
                traffic_source = {"file": "profiles/" + os.path.basename(tpath), "measured_in_run": False,
                                  "kernels": tj.get("kernels_launched"), "ratio_to_algorithmic": tj.get("ratio_to_algorithmic"),
                                  "what": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command (separate runs), "
                                          "corrected as MI355X_MICROARCH.md prescribes"}
            break
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
                       "corpus": "every line generated on its own: std::mt19937_64, seed 20260921 (+1000 per rank), SURVEY.md section 8(d) field "
                                 "distributions (tools/corpus_gen.cpp); all lines distinct, all match",
                       "engine": {1: "tdfa", 2: "nfa"}[info["engine"]], "lines_per_batch": n,
                       "tdfa_states": info["states"], "byte_classes": info["classes"],
                       "lds_table_bytes": info["table_bytes"], "parallelism": "line-shard x%d" % world,
                       "matched_lines_last_batch": matched,
                       # the tables the large-batch kernel stages (LC_TABLE_TDFA_WIDE_BLOB): workgroup size, bytes, and whether
                       # they carry the one-stamp byte-pair table (device_tables.h TP_FORMAT 1)
                       "compact_tables": compact_tables(rx)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernels.split(",")[0].split("<")[0] if kernels else None, "kernels_launched": kernels,
                         "avg_kernel_ms": round(avg_kernel_s * 1e3, 4),
                         "algorithmic_bytes_per_launch": algo_bytes},
            "cpu_baseline": cpu,
            "end_to_end": e2e,
            "configs": extra or None,
            "per_gpu": [{"rank": i, "MBps": round(g["bytes"] / (g["elapsed_us"] / 1e6) / 1e6, 1),
                         "kernel_ms": g["kernel_us_per_step"] / 1e3, "lines": g["lines"], "matched_last": g["matched_last"],
                         "what": "HBM-resident batch per rank (flat by construction: no shared resource); configs[4].per_gpu is the "
                                 "host-fed table"}
                        for i, g in enumerate(per_gpu)],
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------- configs[3]: 64 pipelines
def tenant_pipelines(n_pipelines, group_lines, seed=20260923):
    """64 tenants: delimiter-separated-field regexes with 6..14 capture groups over three delimiters, and for each a pool of
    ~512 B lines (10 % carry one field too many and must fail)."""
    rng = np.random.default_rng(seed)
    delims = [" ", "|", "\t"]
    esc = {"|": r"\|", " ": " ", "\t": r"\t"}
    out = []
    for p in range(n_pipelines):
        ng = int(rng.integers(6, 15))
        d = delims[p % 3]
        cls = {" ": r"[^ ]", "|": r"[^|]", "\t": r"[^\t]"}[d]
        pattern = esc[d].join("(%s*)" % cls for _ in range(ng))
        lines = []
        for _ in range(group_lines):
            nf = ng + (1 if rng.integers(0, 10) == 0 else 0)
            cuts = np.sort(rng.integers(0, 512 - nf, size=nf - 1))
            lens = np.diff(np.concatenate([[0], cuts, [512 - nf + 1]]))
            lines.append(d.join("".join(chr(97 + int(c)) for c in rng.integers(0, 26, size=int(k))) for k in lens).encode())
        out.append((pattern, lines))
    return out


def pack_lines(lines):
    length = np.array([len(s) for s in lines], dtype=np.uint32)
    off = np.zeros(len(lines) + 1, dtype=np.uint32)
    off[1:] = np.cumsum(length.astype(np.uint64) + 1).astype(np.uint32)
    data = np.frombuffer(b"\n".join(lines) + b"\n", dtype=np.uint8).copy()
    return data, off, length


def run_multitenant(args):
    world, rank, dev = setup_dist()
    print(json.dumps(compute_multitenant(args, dev)))


def compute_multitenant(args, dev):
    import torch

    from loongcollector_amd import binding

    P, GL = args.pipelines, args.group_lines
    tenants = tenant_pipelines(P, GL)
    pipes = []
    for pattern, lines in tenants:
        rx = binding.GpuRegex(pattern)
        data, off, length = pack_lines(lines)
        pipes.append({"rx": rx, "pattern": pattern, "data": data, "off": off, "length": length,
                      "d_data": torch.from_numpy(data).to(dev), "d_off": torch.from_numpy(off.view(np.int32)).to(dev),
                      "d_caps": torch.empty((GL, 2 * rx.groups), dtype=torch.int32, device=dev),
                      "d_status": torch.empty((GL,), dtype=torch.uint8, device=dev), "bytes": int(length.sum())})
    streams = [torch.cuda.Stream() for _ in range(args.streams)]

    def turn():  # one round-robin turn: one group per pipeline, as ProcessQueueManager::PopItem hands them out
        for i, p in enumerate(pipes):
            s = streams[i % len(streams)]
            p["rx"].match_device(p["d_data"], p["d_off"], None, GL, p["d_caps"], p["d_status"], sep_bytes=1, stream=s.cuda_stream)

    # the MI355X-native form of the turn: ONE launch whose workgroups each stage their own pipeline's tables (lc_regex_match_device_multi)
    job_array = binding.make_jobs([(p["rx"], p["d_data"], p["d_off"], None, GL, p["d_caps"], p["d_status"], 1) for p in pipes])
    cur = torch.cuda.current_stream().cuda_stream

    def turn_packed():
        binding.match_device_multi(job_array, cur)

    turn_packed()
    torch.cuda.synchronize()
    from oracle.oracle import OracleRegex  # the checker (parity gates below), never the measured path
    if not args.no_cpu_baseline:  # parity gate: every pipeline's group against the oracle (results of the packed launch)
        for p in pipes:
            ec, es = OracleRegex(p["pattern"]).fullmatch_batch(p["data"], p["off"][:-1], p["length"])
            if not (np.array_equal(p["d_status"].cpu().numpy(), es) and np.array_equal(p["d_caps"].cpu().numpy(), ec)):
                raise SystemExit("PARITY FAILURE: pipeline %r differs from the oracle" % p["pattern"])
    for p in pipes:
        p["d_status"].fill_(9)
    turn()
    torch.cuda.synchronize()
    if not args.no_cpu_baseline:  # ... and of the launch-per-group form
        for p in pipes:
            _, es = OracleRegex(p["pattern"]).fullmatch_batch(p["data"], p["off"][:-1], p["length"])
            if not np.array_equal(p["d_status"].cpu().numpy(), es):
                raise SystemExit("PARITY FAILURE: pipeline %r differs from the oracle (launch per group)" % p["pattern"])
    for _ in range(args.warmup):
        turn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        turn()
    torch.cuda.synchronize()
    per_group_elapsed = time.perf_counter() - t0
    for _ in range(args.warmup):
        turn_packed()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        turn_packed()
    ev1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps
    total = sum(p["bytes"] for p in pipes) * args.steps
    # the same bytes with NO tenant switch: pipeline 0's regex over a batch of P groups in one launch
    p0 = pipes[0]
    big = pack_lines(tenants[0][1] * P)
    d_bd, d_bo = torch.from_numpy(big[0]).to(dev), torch.from_numpy(big[1].view(np.int32)).to(dev)
    d_bc = torch.empty((GL * P, 2 * p0["rx"].groups), dtype=torch.int32, device=dev)
    d_bs = torch.empty((GL * P,), dtype=torch.uint8, device=dev)
    cs = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        p0["rx"].match_device(d_bd, d_bo, None, GL * P, d_bc, d_bs, sep_bytes=1, stream=cs)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        p0["rx"].match_device(d_bd, d_bo, None, GL * P, d_bc, d_bs, sep_bytes=1, stream=cs)
    torch.cuda.synchronize()
    single = time.perf_counter() - t1
    launches = P * args.steps
    algo = sum((512 + 5 + 8 * p["rx"].groups) * GL for p in pipes) * args.steps
    # several PENDING groups per pipeline in one packed launch (the process queues hold more than one group per pipeline when the
    # agent is busy, ProcessQueueManager.cpp; a 1000-line group is only 4 workgroups, so 64 of them leave most of the chip waiting)
    pending = {}
    for M in (4, 16):
        caps_m = [torch.empty((M, GL, 2 * p["rx"].groups), dtype=torch.int32, device=dev) for p in pipes]
        stat_m = [torch.full((M, GL), 9, dtype=torch.uint8, device=dev) for p in pipes]
        jobs_m = binding.make_jobs([(p["rx"], p["d_data"], p["d_off"], None, GL, caps_m[i][k], stat_m[i][k], 1)
                                    for k in range(M) for i, p in enumerate(pipes)])
        binding.match_device_multi(jobs_m, cur)
        torch.cuda.synchronize()
        for i, p in enumerate(pipes):  # parity: every copy equals the single-group result checked against the oracle above
            if not (torch.equal(stat_m[i], p["d_status"].expand(M, GL)) and torch.equal(caps_m[i], p["d_caps"].expand(M, GL, -1))):
                raise SystemExit("PARITY FAILURE: %d pending groups per launch differ from the single-group results" % M)
        for _ in range(args.warmup):
            binding.match_device_multi(jobs_m, cur)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            binding.match_device_multi(jobs_m, cur)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.steps
        pending[str(M)] = {"MBps": round(sum(p["bytes"] for p in pipes) * M / (ms * 1e-3) / 1e6, 1), "ms_per_launch": round(ms, 4),
                           "roofline_frac": round(algo / args.steps * M / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
        del caps_m, stat_m, jobs_m
    out = {"metric": "aggregate MB/s parsed, %d pipelines round-robin on 1 MI355X" % P, "value": round(total / elapsed / 1e6, 1),
           "unit": "MB/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": "configs[3]: %d pipelines, each its own regex (6-14 groups), one %d-line (~%d KB) group per pipeline per "
                                  "turn, resident in HBM; a turn = ONE packed launch (lc_regex_match_device_multi: every workgroup "
                                  "stages its own pipeline's tables into LDS)" % (P, GL, 512 * GL >> 10),
                      "launches_per_step": 1, "table_bytes": [p["rx"].info()["table_bytes"] for p in pipes][:8]},
           "launch_per_group": {"what": "the same turn as %d launches round-robin over %d streams" % (P, len(streams)),
                                "MBps": round(total / per_group_elapsed / 1e6, 1),
                                "per_launch_us": round(per_group_elapsed / launches * 1e6, 2)},
           "no_switch": {"what": "the same bytes as ONE launch of pipeline 0's regex over %d lines" % (GL * P),
                         "MBps": round(p0["bytes"] * P * args.steps / single / 1e6, 1)},
           "per_switch_overhead_us": round((elapsed - single) / launches * 1e6, 3),
           "pending_groups_per_pipeline": dict(pending, what="the packed launch with M groups queued per pipeline (64 x M jobs in ONE launch)"),
           "roofline": {"bound": "hbm", "achieved": round(algo / args.steps / (kernel_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": round(algo / args.steps / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5), "traffic": None,
                        "kernel": "tdfa_stream_multi_kernel", "avg_kernel_ms": round(kernel_ms, 4),
                        "algorithmic_bytes_per_launch": algo // args.steps}}
    return out


# ------------------------------------------------------------------------------------------------- configs[4]: host-fed corpus
def run_sharded_corpus(args):
    import torch.distributed as dist
    world, rank, dev = setup_dist()
    out = compute_sharded_corpus(args, world, rank, dev)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def compute_sharded_corpus(args, world, rank, dev):
    """every rank calls this (it contains barriers and the job's all-gather); rank 0 gets the result, the others None"""
    import torch
    import torch.distributed as dist

    from loongcollector_amd import binding, corpus
    from loongcollector_amd.shard import gather_job, run_slab_job

    rx = binding.GpuRegex(corpus.REGEX_B)
    G = rx.groups
    slab_bytes = args.slab_mib << 20
    total_bytes = int(args.corpus_gb * (1 << 30))
    n_slabs_total = max(world, total_bytes // slab_bytes)
    # (64 MiB slabs dealt round-robin to the ranks, SURVEY section 8e: loongcollector_amd/shard.py deal_slabs / run_slab_job)
    # distinct slab contents: a few slabs of mixed nginx/JSON lines (log-uniform 128..2048 B, 30 % JSON that must fail), cycled
    kinds = 4
    slabs = []
    for k in range(kinds):
        data, off, length = corpus.mixed_batch(int(slab_bytes / 640), seed=corpus.SEED + 7 + k)
        nl = int(np.searchsorted(off, slab_bytes, side="right")) - 1  # whole lines only
        nb = int(off[nl])
        h = torch.empty(slab_bytes, dtype=torch.uint8).pin_memory()
        h[:nb] = torch.from_numpy(data[:nb])
        slabs.append({"host": h, "nbytes": nb, "lines": nl, "payload": int(length[:nl].sum()), "data": data, "off": off, "length": length})
    max_lines = max(s["lines"] for s in slabs) + 1
    NBUF = 3
    bufs = []
    for _ in range(NBUF):
        bufs.append({"d_data": torch.empty(slab_bytes + 64, dtype=torch.uint8, device=dev),
                     "d_off": torch.empty(max_lines + 1, dtype=torch.int32, device=dev),
                     "d_n": torch.zeros(1, dtype=torch.int32, device=dev),
                     "d_scratch": torch.empty(binding.split_scratch_bytes(slab_bytes) // 4 + 1, dtype=torch.int32, device=dev),
                     "d_caps": torch.empty((max_lines, 2 * G), dtype=torch.int32, device=dev),
                     "d_status": torch.empty((max_lines,), dtype=torch.uint8, device=dev),
                     "h_caps": torch.empty((max_lines, 2 * G), dtype=torch.int32).pin_memory(),
                     "h_status": torch.empty((max_lines,), dtype=torch.uint8).pin_memory(),
                     "ev": None})
    s_up, s_run, s_down = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    counters = {"bytes": 0, "lines": 0, "matched": 0, "failed": 0}
    timing = []
    done_at = []  # (host clock when the slab's results had arrived, raw bytes of the slab): the steady-state window below

    def feed(slab, buf):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        with torch.cuda.stream(s_up):
            if buf["ev"] is not None:
                s_up.wait_event(buf["ev"][5])  # the buffer's previous results have left
            ev[0].record(s_up)
            buf["d_data"][:slab["nbytes"]].copy_(slab["host"][:slab["nbytes"]], non_blocking=True)
            ev[1].record(s_up)
        with torch.cuda.stream(s_run):
            s_run.wait_event(ev[1])
            ev[2].record(s_run)
            binding.split_lines_device(buf["d_data"], slab["nbytes"], buf["d_off"], buf["d_n"], buf["d_scratch"], stream=s_run.cuda_stream)
            rx.match_device_dyn(buf["d_data"], buf["d_off"], buf["d_n"], max_lines, buf["d_caps"], buf["d_status"], sep_bytes=1,
                                stream=s_run.cuda_stream)
            ev[3].record(s_run)
        with torch.cuda.stream(s_down):
            s_down.wait_event(ev[3])
            ev[4].record(s_down)
            buf["h_caps"][:slab["lines"]].copy_(buf["d_caps"][:slab["lines"]], non_blocking=True)
            buf["h_status"][:slab["lines"]].copy_(buf["d_status"][:slab["lines"]], non_blocking=True)
            ev[5].record(s_down)
        buf["ev"] = ev
        return ev

    def drain(slab, buf, ev):
        ev[5].synchronize()
        st = buf["h_status"][:slab["lines"]].numpy()
        m = int((st == 1).sum())
        counters["bytes"] += slab["payload"]
        counters["lines"] += slab["lines"]
        counters["matched"] += m
        counters["failed"] += slab["lines"] - m
        timing.append((ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3]), ev[4].elapsed_time(ev[5])))
        done_at.append((time.perf_counter(), slab["nbytes"]))

    # parity gate on the first slab (bounded sample of its lines)
    ev = feed(slabs[0], bufs[0])
    ev[5].synchronize()
    if int(bufs[0]["d_n"].item()) != slabs[0]["lines"]:
        raise SystemExit("PARITY FAILURE: split kernel found %d lines, the corpus has %d" % (int(bufs[0]["d_n"].item()), slabs[0]["lines"]))
    if rank == 0 and not args.no_cpu_baseline:
        from oracle.oracle import OracleRegex
        k = min(20000, slabs[0]["lines"])
        ec, es = OracleRegex(corpus.REGEX_B).fullmatch_batch(slabs[0]["data"], slabs[0]["off"][:k], slabs[0]["length"][:k])
        if not (np.array_equal(bufs[0]["h_status"][:k].numpy(), es) and np.array_equal(bufs[0]["h_caps"][:k].numpy(), ec)):
            raise SystemExit("PARITY FAILURE: host-fed slab differs from the oracle")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    my_slabs = run_slab_job(n_slabs_total, rank, world,
                            lambda sidx, b: (slabs[sidx % kinds], bufs[b], feed(slabs[sidx % kinds], bufs[b])),
                            lambda ticket: drain(*ticket), in_flight=NBUF)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    tsum = np.array(timing).sum(axis=0) if timing else np.zeros(3)
    # steady-state window of THIS rank: between the arrival of slab NBUF-1 (the pipeline is full) and of slab n-NBUF-1 (it starts to
    # drain) -- what a long-running agent sees; the whole-run figure beside it carries the fill and the drain of a bounded run
    steady_us, steady_bytes = 0, 0
    if len(done_at) > 2 * NBUF + 2:
        lo, hi = NBUF - 1, len(done_at) - NBUF - 1
        steady_us = int((done_at[hi][0] - done_at[lo][0]) * 1e6)
        steady_bytes = int(sum(b for _, b in done_at[lo + 1:hi + 1]))
    per_gpu = gather_job(dict(counters, elapsed_us=int(elapsed * 1e6), steady_us=steady_us, steady_bytes=steady_bytes,
                              h2d_us=int(tsum[0] * 1e3), kernel_us=int(tsum[1] * 1e3),
                              d2h_us=int(tsum[2] * 1e3), slabs=len(my_slabs), numa_node=PLACEMENT.get("numa_node", -1),
                              cpus=PLACEMENT.get("cpus", 0), pinned=int(bool(PLACEMENT.get("pinned")))), device=dev)
    if rank == 0:
        el = max(g["elapsed_us"] for g in per_gpu) / 1e6
        tot = sum(g["bytes"] for g in per_gpu)
        raw = sum(slabs[s % kinds]["nbytes"] for s in range(n_slabs_total))
        out = {"metric": "aggregate node MB/s parsed, mixed nginx/JSON corpus fed from host", "value": round(tot / el / 1e6, 1),
               "unit": "MB/s", "n_gpus": world, "steps": len(my_slabs), "warmup": 1, "ms_per_step": round(el / max(1, len(my_slabs)) * 1e3, 3),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": "configs[4]: %.1f GB of mixed nginx (regex B, 11 groups) / JSON lines (128-2048 B log-uniform, 30 %% JSON "
                                      "that must fail), %d MiB slabs dealt round-robin to %d rank(s), fed from pinned host memory: "
                                      "H2D -> split kernel -> match kernel -> D2H, %d slabs in flight per rank; the corpus is %d distinct "
                                      "slabs cycled" % (raw / 1e9, args.slab_mib, world, NBUF, kinds),
                          "parallelism": "line-shard x%d, no data-path collective, one all-gather of the counters" % world},
               "pcie": {"raw_bytes_up_GBps": round(raw / el / 1e9, 2), "peak_GBps": PCIE_GEN5_X16_GBPS * world,
                        "frac": round(raw / el / 1e9 / (PCIE_GEN5_X16_GBPS * world), 3),
                        # the window in which every rank's pipeline is full (sum over ranks of bytes / time of each rank's own window)
                        "steady_GBps": round(sum(g["steady_bytes"] / max(1, g["steady_us"]) * 1e6 for g in per_gpu) / 1e9, 2),
                        "steady_frac": round(sum(g["steady_bytes"] / max(1, g["steady_us"]) * 1e6 for g in per_gpu) / 1e9
                                             / (PCIE_GEN5_X16_GBPS * world), 3),
                        "what": "frac: the whole bounded run, fill and drain of the %d-slab pipeline included; steady_frac: between "
                                "the arrival of slab %d and of slab n-%d on each rank" % (NBUF, NBUF - 1, NBUF + 1)},
               "per_gpu": [{"rank": i, "MBps": round(g["bytes"] / (g["elapsed_us"] / 1e6) / 1e6, 1), "lines": g["lines"],
                            "matched": g["matched"], "failed": g["failed"], "slabs": g["slabs"], "h2d_ms": g["h2d_us"] / 1e3,
                            "kernel_ms": g["kernel_us"] / 1e3, "d2h_ms": g["d2h_us"] / 1e3,
                            "placement": {"numa_node": g["numa_node"], "cpus": g["cpus"], "pinned_to_node": bool(g["pinned"])}}
                           for i, g in enumerate(per_gpu)]}
        return out
    return None


def run_launch_check(args):
    """The launch path without a device (CPU test of `--gpus N`): rendezvous, ONE all-gather of a counter row (the job's only
    collective, shard.gather_job), rank 0 prints the launch fields of the line.  No parsing happens and the line says so."""
    import torch.distributed as dist
    from loongcollector_amd.shard import gather_job
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group(backend="gloo")
    per_gpu = gather_job({"bytes": 0, "lines": 0, "rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
                          "elapsed_us": 1}, device="cpu")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"launch_check": True, "value": None, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "data": "none (launch path only: no device work, no parsing)",
                          "per_gpu": [{"rank": g["rank"], "local_rank": g["local_rank"]} for g in per_gpu]}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="2: the headline (BASELINE configs[1]); 3: Grok, 50 patterns (configs[2], = tools/grok_bench.py); "
                         "4: multi-tenant (configs[3]); 5: host-fed sharded corpus (configs[4])")
    ap.add_argument("--lines", type=int, default=1 << 20, help="lines per batch per GPU")
    ap.add_argument("--line-bytes", type=int, default=512)
    ap.add_argument("--regex", choices=["A", "B"], default="A", help="A: 10-group doc regex, B: 11-group benchmark regex")
    ap.add_argument("--engine", choices=["auto", "tdfa", "nfa"], default="auto")
    ap.add_argument("--cpu-sample-lines", type=int, default=1 << 20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-inclusive measurements (end_to_end)")
    ap.add_argument("--no-configs", action="store_true", help="skip the bounded runs of the other BASELINE configs (configs)")
    ap.add_argument("--pipelines", type=int, default=64)
    ap.add_argument("--group-lines", type=int, default=1000)
    ap.add_argument("--streams", type=int, default=4)
    ap.add_argument("--corpus-gb", type=float, default=10.0)
    ap.add_argument("--slab-mib", type=int, default=64)
    ap.add_argument("--launch-check", action="store_true",
                    help="(tests) no device work: the ranks rendezvous over gloo, all-gather a fixed counter row and rank 0 prints the "
                         "line's launch fields -- what tests/test_bench_launch.py runs on a box without a GPU")
    args = ap.parse_args()
    # one process per GPU: under a launcher WORLD_SIZE must equal --gpus; without one, --gpus N > 1 starts the N ranks itself
    from loongcollector_amd.launch import ensure_ranks
    ensure_ranks(args.gpus, need_devices=not args.launch_check)
    if args.launch_check:
        return run_launch_check(args)
    if args.config == 3:  # the Grok line has its own driver (parity gate against the Grok oracle, per-pattern engines)
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
        import grok_bench
        lines = args.lines if "--lines" in sys.argv else 1 << 18
        sys.argv = [sys.argv[0], "--lines", str(lines), "--steps", str(min(args.steps, 5)), "--warmup", str(max(args.warmup, 4))]
        grok_bench.main()
    elif args.config == 4:
        run_multitenant(args)
    elif args.config == 5:
        run_sharded_corpus(args)
    else:
        run_headline(args)


if __name__ == "__main__":
    main()
