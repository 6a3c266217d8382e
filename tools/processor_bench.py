#!/usr/bin/env python3
"""In-agent shaped number: lc_processor_process (gather views -> one GPU match -> zero-copy stitch + policy) on event
groups of ~1000 lines (what ProcessorRunner hands over, ProcessorRunner.cpp:138-142) and on one large group.
Single host thread.  Quoted in DESIGN.md next to the kernel-only headline; never the headline itself."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loongcollector_amd import corpus  # noqa: E402
from loongcollector_amd.processor import EventGroup, Processor  # noqa: E402

n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
data, off, length = corpus.apache_batch(n_total, "A")
raw = data.tobytes()
lines = [raw[off[i]:off[i] + length[i]].decode("latin-1") for i in range(n_total)]
cfg = {"SourceKey": "content", "Regex": corpus.REGEX_A, "Keys": corpus.KEYS_A}
for group_lines in (1000, n_total):
    groups = [EventGroup({"events": [{"contents": {"content": s}, "timestamp": 1, "type": 1} for s in lines[i:i + group_lines]]})
              for i in range(0, n_total, group_lines)]
    p = Processor(cfg)
    p.process(EventGroup({"events": [{"contents": {"content": lines[0]}, "timestamp": 1, "type": 1}]}))  # warm-up
    t0 = time.perf_counter()
    for g in groups:
        p.process(g)
    dt = time.perf_counter() - t0
    c = p.counters()
    assert c["out_successful_events_total"] == n_total + 1
    print("processor: groups of %7d lines: %.1f ms total, %.2f us/line, %.1f MB/s parsed (gather + H2D + kernel + D2H + stitch, 1 thread)"
          % (group_lines, dt * 1e3, dt / n_total * 1e6, n_total * 512 / dt / 1e6))
