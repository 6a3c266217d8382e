#!/usr/bin/env python3
"""BASELINE configs[2] as bench.py embeds it: the Grok batches (tools/grok_bench.py) and the in-agent shape
(tools/grok_inagent_bench.py) in ONE process -- its own, because a process that hosts a Grok processor runs with 16 hardware
queues (GPU_MAX_HW_QUEUES; csrc/gpu_runtime.hip) and the HIP runtime reads that when it initialises.  Prints one JSON object."""
import argparse
import json
import os
import sys

# The HOST's decision (the library never touches the environment: include/lc_regex_gpu.h lc_runtime_prefer_hw_queues): a process that
# hosts Grok processors asks for 16 hardware queues, before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--lines", default="16384,65536")
    args = ap.parse_args()
    import grok_bench
    import grok_inagent_bench
    # (parity gate: 3 000 lines strided across every batch through the oracle before anything is timed -- ~10 s per batch size)
    g = grok_bench.measure(argparse.Namespace(lines=args.lines, steps=10, warmup=4, cpu_sample_lines=3000, patterns=0,
                                              no_sequential_check=False, sequential=False), device_index=args.device)
    for r in g:
        r["config"].pop("patterns_refused", None)
    out = {"batches": g,
           "in_agent": grok_inagent_bench.measure(argparse.Namespace(threads="1,16", group=1000, groups=20, patterns=0)),
           "process": {"what": "tools/grok_config2.py, a process of its own", "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
