#!/usr/bin/env python3
"""PCIe-inclusive rate of the host entry points (lc_regex_match_host / _views): pinned double-buffered staging,
H2D + kernel + D2H overlapped on two streams.  Never the headline `value` (that is HBM-resident); quoted in DESIGN.md."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loongcollector_amd import binding, corpus  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
data, off, length = corpus.apache_batch(n, "A")
rx = binding.GpuRegex(corpus.REGEX_A)
rx.match_host(data, off[:-1], length)  # warm-up: allocates pinned + device staging
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    caps, status = rx.match_host(data, off[:-1], length)
    best = min(best, time.perf_counter() - t0)
assert status.all()
print("match_host: %d lines, %.1f MB payload, best %.2f ms -> %.2f GB/s parsed (PCIe-inclusive, 1 host thread gathers)"
      % (n, length.sum() / 1e6, best * 1e3, length.sum() / best / 1e9))
