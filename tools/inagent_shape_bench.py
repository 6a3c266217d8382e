#!/usr/bin/env python3
"""bench.py's in-agent leg (lc_processor_process on 1000-line event groups, N runner threads, 16 groups alive per thread) in a process
of its own, so that it can run on ANOTHER build of the library: LC_REGEX_GPU_LIB selects it.  bench.py starts this with
loongcollector_amd/lib/liblc_regex_gpu_refshape.so -- the stand-in event model built in the reference's shape (heap std::vector
contents reserved to 16, no chunk pool, K x SetContentNoCopy + DelContent: csrc/event_model.hpp LC_REFERENCE_SHAPED_EVENT_MODEL) --
for end_to_end.in_agent_reference_shape_MBps: what one runner thread does with the event model an agent build has.
Every group's first event is checked against the oracle's captures by the caller's parity gate form (fields of line 0).
Usage: inagent_shape_bench.py --threads 1,16,32 [--lines 262144] [--group-lines 1000] [--regex A]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="1")
    ap.add_argument("--lines", type=int, default=1 << 18)
    ap.add_argument("--group-lines", type=int, default=1000)
    ap.add_argument("--regex", choices=["A", "B"], default="A")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from loongcollector_amd import binding, corpus
    from oracle.oracle import OracleRegex
    torch.cuda.set_device(args.device)
    binding.set_bind_policy(binding.LC_BIND_FIXED, args.device)
    pattern = corpus.REGEX_A if args.regex == "A" else corpus.REGEX_B
    keys = corpus.KEYS_A if args.regex == "A" else corpus.KEYS_B
    data, off, length = corpus.apache_batch(args.lines, args.regex)
    m = args.lines // args.group_lines * args.group_lines
    exp_caps, exp_status = OracleRegex(pattern).fullmatch_batch(data, off[:1], length[:1])
    raw = data[int(off[0]):int(off[0]) + int(length[0])].tobytes()
    want = [(k, raw[exp_caps[0][2 * i]:exp_caps[0][2 * i + 1]].decode("latin-1")) for i, k in enumerate(keys)]
    out = {"lib": os.path.basename(binding.LIB_PATH), "window16_MBps": {}, "all_groups_alive_MBps": {}}
    out["window16_one_thread_runs_MBps"] = []
    for t in [int(x) for x in args.threads.split(",")]:
        # one runner thread on this event model moves 2.3-4.5 GB/s between runs (heap behaviour of 1000 x std::vector per group): the
        # figure in the line is the MEDIAN of five repetitions in this process, the five are listed beside it
        reps = 5 if t == 1 else 1
        runs = []
        for _ in range(reps):
            mbps, first = bench.measure_in_agent_window(pattern, keys, data, off, length, args.group_lines, m // args.group_lines, t)
            if [tuple(kv) for kv in first] != want:
                raise SystemExit("PARITY FAILURE (in-agent path, %s): stitched fields differ from the oracle's captures" % out["lib"])
            runs.append(round(mbps, 1))
        out["window16_MBps"][str(t)] = sorted(runs)[len(runs) // 2]
        if t == 1:
            out["window16_one_thread_runs_MBps"] = runs
    t0 = int(args.threads.split(",")[0])
    out["all_groups_alive_MBps"][str(t0)] = round(bench.measure_in_agent(pattern, keys, data, off, length, args.group_lines, m // args.group_lines, t0)[0], 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
