#!/usr/bin/env python3
"""Per-pattern timing of the Grok regex kernels on the configs[2] corpus: for a few Match patterns, the search kernel over
(a) the lines the pattern matches and (b) every line, with line-length buckets.  Diagnostic tool."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from loongcollector_amd import binding as B  # noqa: E402
from loongcollector_amd.grok import Grok  # noqa: E402
from loongcollector_amd.grok_corpus import grok_lines  # noqa: E402

F = B.LC_SYNTAX_SEARCH | B.LC_SYNTAX_NAMED_ONLY | B.LC_SYNTAX_NO_DOTALL | B.LC_SYNTAX_NO_MULTILINE | B.LC_SYNTAX_REGEXP2
cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "grok_config3.json"), encoding="utf-8"))
# LC_BENCH_ANCHORED=1: the anchored search (LC_SYNTAX_SEARCH | LC_SYNTAX_PREFIX) -- what the Grok matcher's round 0 launches for an entry
# LC_BENCH_ENGINE=nfa: force the thread-list engine (the entries whose tagged DFA does not build run on it anyway)
if os.environ.get("LC_BENCH_ANCHORED"):
    F |= B.LC_SYNTAX_PREFIX
ENGINE = B.LC_ENGINE_NFA if os.environ.get("LC_BENCH_ENGINE") == "nfa" else B.LC_ENGINE_AUTO
REPS = int(os.environ.get("LC_BENCH_REPS", "3"))
names = sys.argv[1:] or ["%{CATALINALOG}", "%{TOMCATLOG}", "%{CISCOFW106001}", "%{SYSLOGLINE}", "%{CRONLOG}"]
lib = Grok(CustomPatterns=cfg["custom_patterns"])
lines = grok_lines(8192)
dev = torch.device("cuda:0")


def run(rx, subset, label):
    n = len(subset)
    if not n:
        return
    length = np.array([len(v) for v in subset], dtype=np.uint32)
    off = np.zeros(n, dtype=np.uint32)
    off[1:] = np.cumsum(length[:-1], dtype=np.uint64).astype(np.uint32)
    data = np.frombuffer(b"".join(subset) + b"\0" * 16, dtype=np.uint8).copy()
    d = torch.from_numpy(data).to(dev)
    o = torch.from_numpy(off.view(np.int32)).to(dev)
    l = torch.from_numpy(length.view(np.int32)).to(dev)
    G = rx.groups
    caps = torch.empty((n, 2 * G), dtype=torch.int32, device=dev)
    st = torch.empty(n, dtype=torch.uint8, device=dev)
    rx.match_device(d, o, l, n, caps, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS):
        rx.match_device(d, o, l, n, caps, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / REPS
    s = st.cpu().numpy()
    print("  %-22s n=%5d bytes=%8d max=%4d  %.3f ms  %.2f us/line  matched=%d overflow=%d"
          % (label, n, int(length.sum()), int(length.max()), dt * 1e3, dt / n * 1e6, int((s == 1).sum()), int((s == 2).sum())))


for name in names:
    rx = B.GpuRegex(lib.denormalize(name).encode("utf-8"), syntax_flags=F, engine=ENGINE)
    if os.environ.get("LC_BENCH_PREFER_WAVE"):   # what the Grok matcher asks for its entries: one value per wavefront on small batches
        rx.prefer_wave_tdfa()
    info = rx.info()
    lit = rx.required_literal()
    print(name, "engine", info["engine"], "table_bytes", info["table_bytes"], "groups", rx.groups, "literal", lit, "kernels", B.launched_kernels())
    hits = [v for v in lines if lit in v] if lit else lines
    run(rx, hits, "lines with literal")
    run(rx, [v for v in hits if len(v) <= 256], "  of those <= 256 B")
    run(rx, [v for v in hits if len(v) >= 2048], "  of those >= 2 KiB")
    run(rx, lines, "all lines")
    print("  kernels:", B.launched_kernels())
